from fvcore.common.config import CfgNode  # noqa
