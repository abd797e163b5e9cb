"""Mirror of hloc/utils/base_model.py (reference): the plugin boundary of the hloc
matchers.  BaseModel merges default_conf with the given conf, checks required keys
and dispatches to _forward; dynamic_load picks the single BaseModel subclass of a
module (hloc/utils/base_model.py:40-49).  No nn.Module behind it: .eval()/.to()/.cuda()
are accepted because the drivers call them (hloc/match_features.py:78-79)."""
import inspect
from abc import ABCMeta, abstractmethod
from copy import copy


class BaseModel(metaclass=ABCMeta):
    default_conf = {}
    required_data_keys = []

    def __init__(self, conf):
        self.conf = conf = {**self.default_conf, **conf}
        self.required_data_keys = copy(self.required_data_keys)
        self._device = 0
        self._init(conf)

    def forward(self, data):
        for key in self.required_data_keys:
            assert key in data, 'Missing key {} in data'.format(key)
        return self._forward(data)

    __call__ = forward

    def eval(self):
        return self

    def cuda(self, device=None):
        return self

    def to(self, device):
        if isinstance(device, str) and not device.startswith("cuda"):
            raise RuntimeError("sfd2_amd runs on the MI355X only; there is no CPU path")
        return self

    @abstractmethod
    def _init(self, conf):
        raise NotImplementedError

    @abstractmethod
    def _forward(self, data):
        raise NotImplementedError


def dynamic_load(root, model):
    module_path = f'{root.__name__}.{model}'
    module = __import__(module_path, fromlist=[''])
    classes = inspect.getmembers(module, inspect.isclass)
    classes = [c for c in classes if c[1].__module__ == module_path]
    classes = [c for c in classes if issubclass(c[1], BaseModel)]
    assert len(classes) == 1, classes
    return classes[0][1]
