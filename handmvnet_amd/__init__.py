"""handmvnet_amd -- MI355X-native (gfx950) HandMvNet inference forward pass.

`from handmvnet_amd import HandMvNet` is the drop-in for the reference's
`from models.handmvnet import HandMvNet` on the inference path.
"""
from .spec import HotPathConfig, config_from_params, state_dict_layout  # noqa: F401


def __getattr__(name):
    if name == "HandMvNet":  # lazy: importing the package must not require torch.cuda / the .so
        from .model import HandMvNet
        return HandMvNet
    raise AttributeError(name)
