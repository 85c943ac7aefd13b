"""EventStorage shim (detectron2/utils/events.py:232-431).  The reference calls
`get_event_storage().put_scalar(...)` from inside the hot path and floats the value at once — one
device->host sync per call.  Here scalars are kept as they come (device tensors
stay on the device) and are only materialised when somebody reads them."""
from collections import defaultdict
from contextlib import contextmanager

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


def has_event_storage():
    return len(_CURRENT_STORAGE_STACK) > 0


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(list)
        self._latest = {}
        self._iter = start_iter
        self._current_prefix = ""

    def put_scalar(self, name, value, smoothing_hint=True):
        name = self._current_prefix + name
        self._history[name].append((value, self._iter))
        self._latest[name] = value

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint)

    def latest(self):
        return {k: float(v) for k, v in self._latest.items()}  # sync happens here, on demand

    def history(self, name):
        return [(float(v), it) for v, it in self._history[name]]

    def step(self):
        self._iter += 1

    @property
    def iter(self):
        return self._iter

    @iter.setter
    def iter(self, val):
        self._iter = int(val)

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        assert _CURRENT_STORAGE_STACK[-1] == self
        _CURRENT_STORAGE_STACK.pop()

    @contextmanager
    def name_scope(self, name):
        old = self._current_prefix
        self._current_prefix = name.rstrip("/") + "/"
        yield
        self._current_prefix = old
