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
    """The plugin lookup of the hloc drivers (hloc/utils/base_model.py:40-49, called from hloc/match_features.py:78):
    `root` is a package (e.g. sfd2_amd.matchers), `model` the name of one of its modules; the module must define exactly
    one BaseModel subclass of its own, which is returned."""
    import importlib
    name = root.__name__ + "." + model
    mod = importlib.import_module(name)
    own = [obj for obj in vars(mod).values()
           if inspect.isclass(obj) and obj.__module__ == name and issubclass(obj, BaseModel)]
    if len(own) != 1:
        raise AssertionError(f"{name} must define exactly one BaseModel subclass, found {[c.__name__ for c in own]}")
    return own[0]
