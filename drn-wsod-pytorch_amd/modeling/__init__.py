from .backbone import (Backbone, BasicBlock, BasicStem, BottleneckBlock, PlainBlock, ResNet, VGG16, build_backbone,
                       build_vgg_backbone, build_ws_resnet_backbone)
from .rcnn import GeneralizedRCNNWSL, build_model, detector_postprocess
from .roi_heads import (Box2BoxTransform, DiscriminativeAdaptionNeck, Matcher, OICROutputLayers, OICRROIHeads, ROIHeads,
                        PCLROIHeads, ROIPooler, WSDDNOutputLayers, WSDDNROIHeads, build_box_head, build_roi_heads)
from .tta import DatasetMapperTTAAVG, GeneralizedRCNNWithTTAAVG
from ..registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
