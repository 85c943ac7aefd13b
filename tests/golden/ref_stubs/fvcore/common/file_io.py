import os
from contextlib import contextmanager


class PathHandler:
    pass


class _PM:
    def open(self, path, mode="r", **kw):
        return open(path, mode)

    def get_local_path(self, path, **kw):
        return path

    def exists(self, p):
        return os.path.exists(p)

    def isfile(self, p):
        return os.path.isfile(p)

    def isdir(self, p):
        return os.path.isdir(p)

    def ls(self, p):
        return os.listdir(p)

    def mkdirs(self, p):
        os.makedirs(p, exist_ok=True)

    def register_handler(self, h):
        pass


PathManager = _PM()


@contextmanager
def file_lock(path):
    yield
