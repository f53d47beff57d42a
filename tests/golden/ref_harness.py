"""Import the REAL reference (/root/reference, read-only) in the build container.

Only the fixture generator (make_fixtures.py) and ad-hoc validation scripts use this
module; nothing under `-m gpu`, smoke() or bench.py may (the reference does not exist on
the GPU box).  Absent third-party packages are replaced by empty stub modules exactly as
SURVEY.md section 8(c) records: torchvision (weight enums only), cv2, lightning, manopth,
transforms3d.  `backbone_pretrained` is always forced to False (it is a network fetch).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(REFERENCE_SRC)


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns the reference's HandMvNet class."""
    import torch

    sys.dont_write_bytecode = True
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    if "models.handmvnet" in sys.modules:
        return sys.modules["models.handmvnet"].HandMvNet

    class _Weights:
        class DEFAULT:
            url = ""

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tvm = _stub("torchvision.models")
        tvr = _stub("torchvision.models.resnet", ResNet18_Weights=_Weights, ResNet34_Weights=_Weights,
                    ResNet50_Weights=_Weights)
        tv.models, tvm.resnet = tvm, tvr
    for name in ("cv2", "transforms3d"):
        if name not in sys.modules:
            _stub(name)
    if "manopth" not in sys.modules:
        mp = _stub("manopth")
        mp.manolayer = _stub("manopth.manolayer", ManoLayer=object)

    if "lightning" not in sys.modules:
        class LightningModule(torch.nn.Module):
            def save_hyperparameters(self, *a, **k):
                pass

            def log(self, *a, **k):
                pass

            def freeze(self):
                for p in self.parameters():
                    p.requires_grad = False
                self.eval()

        _stub("lightning", LightningModule=LightningModule)

    from models.handmvnet import HandMvNet  # noqa: E402
    return HandMvNet


def build_reference_model(train_params: dict, model_params: dict, data_params: dict, state_dict_np):
    """Instantiate the reference model (CPU, eval) and strictly load numpy weights."""
    import torch

    HandMvNet = import_reference()
    mp = dict(model_params)
    mp["backbone_pretrained"] = False
    model = HandMvNet(train_params, mp, data_params).eval()
    sd = {k: torch.from_numpy(v.copy()) if v.ndim else torch.tensor(int(v)) for k, v in state_dict_np.items()}
    model.load_state_dict(sd, strict=True)
    model.freeze()
    if mp.get("fusion") == "cross_attn_learnable_query":
        # HandMvNet.forward (handmvnet.py:227) calls joints_late_fusion(x, add_pos=...), a keyword that
        # CrossAttentionFusionLearnableQuery.forward(self, x) (fusion.py:47) does not take: the reference raises TypeError.
        # The fixtures pin the module itself: same module, same weights, called the only way its signature allows.
        lq = model.joints_late_fusion
        orig = lq.forward
        lq.forward = lambda x, add_pos=True: orig(x)
    return model
