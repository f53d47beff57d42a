"""Host-side description of the HandMvNet hot path: config handling and the
``state_dict`` key/shape layout the engine ingests.

Mirrors (does not import) the reference:
  * constructor keys            -- /root/reference/src/models/handmvnet.py:28-125
  * backbone module tree        -- /root/reference/src/models/backbones/resnet.py:147-203, 312-357
  * pose_net / sample_nets      -- handmvnet.py:70-97, layers.py:318-334, nets.py:24-31
  * fusion blocks               -- fusion.py:7-24, layers.py:177-200, 161-171
  * decoders                    -- nets.py:119-154, layers.py:363-385
  * config derivation           -- /root/reference/src/config.py:44-49 (num_views = len(selected_views))
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

POS2D, CROP, SIN = 1, 2, 4
BACKBONE_IDS = {"18": 0, "34": 1, "50_paper": 2, "w40": 3, "w64": 4}
HRNET_CHANNELS = {"w40": [40, 80, 160, 320], "w64": [64, 128, 256, 512]}   # hrnet.py:430-447
RESNET_BLOCKS = {"18": [2, 2, 2, 2], "34": [3, 4, 6, 3], "50_paper": [3, 4, 6, 3]}
N_JOINTS = 21
N_HEADS = 8
DIM_HEAD = 128
DIM_HEAD_LQ = 256   # MultiHeadAttentionLearnableQuery (layers.py:241)


@dataclass
class HotPathConfig:
    """Everything the engine needs to know to build the forward plan."""
    backbone_type: str = "50_paper"
    backbone_channels: List[int] = field(default_factory=lambda: [1024])
    num_views: int = 8
    image_size: int = 256          # data.image_size (config constant, handmvnet.py:252)
    heatmap_size: int = 32         # data.heatmap_size
    pos_enc: Tuple[str, ...] = ("pos2d", "sin")
    fusion_layers: int = 5
    use_gcn: bool = True
    freeze_bn: bool = False
    early_return: int = 3
    fusion: str = "cross_attn"     # model.fusion: "cross_attn" (every release config) | "cross_attn_learnable_query"

    @property
    def learnable_query(self) -> bool:
        return self.fusion == "cross_attn_learnable_query"

    @property
    def pos_mask(self) -> int:
        return (POS2D if "pos2d" in self.pos_enc else 0) | (CROP if "crop" in self.pos_enc else 0) | \
               (SIN if "sin" in self.pos_enc else 0)

    @property
    def feat_dim(self) -> int:
        # handmvnet.py:88-95
        d = int(sum(self.backbone_channels) / 2)
        if "pos2d" in self.pos_enc:
            d += 2
        if "crop" in self.pos_enc:
            d += 10
        return d

    @property
    def is_paper(self) -> bool:
        return "paper" in self.backbone_type

    @property
    def is_hrnet(self) -> bool:
        return self.backbone_type in HRNET_CHANNELS


def config_from_params(train_params: dict, model_params: dict, data_params: dict) -> HotPathConfig:
    """Validate the three reference dicts exactly where the reference does
    (handmvnet.py:35-125) and distil them into a HotPathConfig."""
    _ = train_params["debug"]
    if not train_params["root_relative"]:
        # handmvnet.py:102-103,236-249 -- the root-joint branch is shape-inconsistent in the
        # reference and no shipped config enables it.
        raise NotImplementedError("root_relative=False is outside the accelerated hot path")
    backbone = model_params.get("backbone", "hrnet")
    assert backbone in ["hrnet", "resnet"], "Backbone should be one of ['hrnet', 'resnet']"
    if backbone == "hrnet":
        btype = model_params.get("backbone_type", "w40")   # handmvnet.py:42
        if btype not in HRNET_CHANNELS:                     # hrnet.py:449
            raise Exception("HRNet only supports ['w64', 'w40'] as model_type, found: " + str(btype))
    else:
        btype = model_params.get("backbone_type", "34")
        assert btype in ["18", "34", "50_paper"], "Supports only 18, 34, 50_paper"
    if model_params["fusion"] not in ("cross_attn", "cross_attn_learnable_query"):   # handmvnet.py:139-149
        raise NotImplementedError(f"Invalid fusion type: {model_params['fusion']}")
    ds_name = data_params.get("name", "dexycb")
    if ds_name not in ("dexycb", "ho3d", "mvhand"):
        raise NotImplementedError(f"Dataset not found: {ds_name}")
    _ = data_params["batch_size"]
    layers = model_params.get("fusion_layers", 5)
    assert layers % 2 == 1, "num_layers must be an odd number"
    if model_params["fusion"] == "cross_attn_learnable_query":
        # An EXTENSION beyond the reference's runnable surface: HandMvNet.forward raises TypeError for this
        # fusion (handmvnet.py:227 passes add_pos=, fusion.py:47 does not take it).  The module itself always has 5 blocks and adds
        # its own positional encoding in every block, so two keys the reference reads have no effect here: say so.
        ignored = []
        if "sin" in model_params.get("pos_enc", ["pos2d", "sin"]):
            ignored.append("'sin' in pos_enc (every learnable-query block adds its own PositionalEncoding, layers.py:273-275)")
        if int(layers) != 5:
            ignored.append(f"fusion_layers={layers} (CrossAttentionFusionLearnableQuery always builds 5 blocks, fusion.py:37-45)")
        if ignored:
            import warnings
            warnings.warn("fusion='cross_attn_learnable_query' ignores " + " and ".join(ignored), stacklevel=2)
    if "num_views" in model_params:
        num_views = int(model_params["num_views"])
    else:  # config.py:46-49
        num_views = len(model_params["selected_views"])
    return HotPathConfig(
        backbone_type=btype,
        backbone_channels=[int(c) for c in model_params["backbone_channels"]],
        num_views=num_views,
        image_size=int(data_params["image_size"]),
        heatmap_size=int(data_params["heatmap_size"]),
        pos_enc=tuple(model_params.get("pos_enc", ["pos2d", "sin"])),
        fusion_layers=int(layers),
        use_gcn=bool(model_params["use_gcn"]),
        # ResNet50_Paper hard-codes freeze_batchnorm=False (resnet.py:354)
        freeze_bn=bool(model_params.get("freeze_bn", False)) and btype in ("18", "34"),
        early_return=int(model_params.get("backbone_early_return", 3)),
        fusion=str(model_params["fusion"]),
    )


def _down(n: int, k: int, s: int, p: int) -> int:
    return (n + 2 * p - k) // s + 1


def level_sizes(cfg: HotPathConfig, h: int, w: int):
    """[(h, w)] of the sampled feature levels in the reference's feats[] order for an h x w frame, following the conv
    arithmetic of the backbones (resnet.py:216-254: 7x7 s2 p3, maxpool 3x3 s2 p1, 3x3 / 1x1 s2 convs -- any frame size;
    hrnet.py:357-393: two 3x3 s2 p1 stem convs, then one more 3x3 s2 p1 per lower branch)."""
    if cfg.is_hrnet:
        h, w = _down(_down(h, 3, 2, 1), 3, 2, 1), _down(_down(w, 3, 2, 1), 3, 2, 1)
        out = [(h, w)]
        for _ in range(3):
            h, w = _down(h, 3, 2, 1), _down(w, 3, 2, 1)
            out.append((h, w))
        return out[:max(len(cfg.backbone_channels), 1)]
    h, w = _down(_down(h, 7, 2, 3), 3, 2, 1), _down(_down(w, 7, 2, 3), 3, 2, 1)   # conv1 + maxpool
    levels = [(h, w)]                                                              # layer1
    h, w = _down(h, 3, 2, 1), _down(w, 3, 2, 1)
    levels.append((h, w))                                                          # layer2
    if not cfg.is_paper:
        h, w = _down(h, 3, 2, 1), _down(w, 3, 2, 1)
    levels.append((h, w))                                                          # layer3 (stride 1 for 50_paper)
    return list(reversed(levels))[:max(len(cfg.backbone_channels), 1)]


def heatmap_size_of(cfg: HotPathConfig, h: int, w: int):
    """(h, w) of joint_hms for an h x w frame (handmvnet.py:180): r50-paper keeps feats[0]'s size, r18/34 double it
    (ConvTranspose2d 4x4 s2 p1), HRNet halves its highest-resolution branch (3x3 s2 p1 conv)."""
    fh, fw = level_sizes(cfg, h, w)[0]
    if cfg.is_hrnet:
        return _down(fh, 3, 2, 1), _down(fw, 3, 2, 1)
    if cfg.is_paper:
        return fh, fw
    return 2 * fh, 2 * fw


def _bn(keys: "OrderedDict[str, tuple]", prefix: str, c: int, frozen: bool) -> None:
    keys[prefix + ".weight"] = (c,)
    keys[prefix + ".bias"] = (c,)
    keys[prefix + ".running_mean"] = (c,)
    keys[prefix + ".running_var"] = (c,)
    if not frozen:
        keys[prefix + ".num_batches_tracked"] = ()


def state_dict_layout(cfg: HotPathConfig) -> "OrderedDict[str, tuple]":
    """Key -> shape, in the reference's module registration order.  Includes the
    never-executed layer4 / fc parameters r18/r34 checkpoints carry."""
    k: "OrderedDict[str, tuple]" = OrderedDict()
    if cfg.is_hrnet:
        _hrnet_layout(k, cfg)
        _heads_layout(k, cfg)
        return k
    fz = cfg.freeze_bn
    blocks = RESNET_BLOCKS[cfg.backbone_type]
    bottleneck = cfg.is_paper
    exp = 4 if bottleneck else 1
    k["backbone.conv1.weight"] = (64, 3, 7, 7)
    _bn(k, "backbone.bn1", 64, fz)
    inplanes = 64
    n_layers = 3 if cfg.is_paper else 4
    for li in range(n_layers):
        planes = 64 << li
        stride = 1 if li == 0 else 2
        if cfg.is_paper and li == 2:
            stride = 1  # resnet.py:176-177
        for bi in range(blocks[li]):
            p = f"backbone.layer{li + 1}.{bi}"
            s = stride if bi == 0 else 1
            if bottleneck:
                k[p + ".conv1.weight"] = (planes, inplanes, 1, 1)
                _bn(k, p + ".bn1", planes, fz)
                k[p + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(k, p + ".bn2", planes, fz)
                k[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
                _bn(k, p + ".bn3", planes * 4, fz)
            else:
                k[p + ".conv1.weight"] = (planes, inplanes, 3, 3)
                _bn(k, p + ".bn1", planes, fz)
                k[p + ".conv2.weight"] = (planes, planes, 3, 3)
                _bn(k, p + ".bn2", planes, fz)
            if bi == 0 and (s != 1 or inplanes != planes * exp):
                k[p + ".downsample.0.weight"] = (planes * exp, inplanes, 1, 1)
                _bn(k, p + ".downsample.1", planes * exp, fz)
            inplanes = planes * exp
    if not cfg.is_paper:
        k["backbone.fc.weight"] = (1000, 512 * exp)
        k["backbone.fc.bias"] = (1000,)
    _heads_layout(k, cfg)
    return k


def _heads_layout(k: "OrderedDict[str, tuple]", cfg: HotPathConfig) -> None:
    """pose_net, sample_nets, fusion and decoder keys (handmvnet.py:46-100)."""
    c0 = cfg.backbone_channels[0]
    if cfg.is_hrnet:   # handmvnet.py:51-57: nn.Conv2d(C0, 21, 3, stride 2, padding 1)
        k["pose_net.weight"] = (21, c0, 3, 3)
        k["pose_net.bias"] = (21,)
    elif cfg.is_paper:
        k["pose_net.0.weight"] = (512, c0, 1, 1)
        k["pose_net.0.bias"] = (512,)
        _bn(k, "pose_net.1", 512, False)
        k["pose_net.3.weight"] = (21, 512, 1, 1)
        k["pose_net.3.bias"] = (21,)
    else:
        k["pose_net.0.weight"] = (c0, 128, 4, 4)  # ConvTranspose2d layout [in, out, kh, kw]
        k["pose_net.0.bias"] = (128,)
        _bn(k, "pose_net.1", 128, False)
        k["pose_net.3.weight"] = (64, 128, 3, 3)
        k["pose_net.3.bias"] = (64,)
        _bn(k, "pose_net.4", 64, False)
        k["pose_net.6.weight"] = (21, 64, 3, 3)
        k["pose_net.6.bias"] = (21,)
    for i, c in enumerate(cfg.backbone_channels):
        k[f"sample_nets.{i}.conv.0.weight"] = (c // 2, c, 1, 1)
        k[f"sample_nets.{i}.conv.0.bias"] = (c // 2,)
        _bn(k, f"sample_nets.{i}.conv.1", c // 2, False)
    d = cfg.feat_dim
    inner = N_HEADS * DIM_HEAD
    if cfg.learnable_query:
        # CrossAttentionFusionLearnableQuery (fusion.py:33-49): always 5 MultiHeadAttentionLearnableQuery blocks
        # (layers.py:240-301): heads 8 x 256, to_out = Sequential(Linear, Dropout), FeedForward hidden 256, no LayerNorm
        # around the attention; the middle block owns the learnable 21-token probe.  pos_embed holds no parameters.
        inner_lq = N_HEADS * DIM_HEAD_LQ
        for l in range(5):
            p = f"joints_late_fusion.attn_fusion.{l}"
            if l == 2:
                k[p + ".probe"] = (1, N_JOINTS, d)
            k[p + ".to_q.weight"] = (inner_lq, d)
            k[p + ".to_k.weight"] = (inner_lq, d)
            k[p + ".to_v.weight"] = (inner_lq, d)
            k[p + ".to_out.0.weight"] = (d, inner_lq)
            k[p + ".to_out.0.bias"] = (d,)
            k[p + ".ff.net.0.weight"] = (d,)
            k[p + ".ff.net.0.bias"] = (d,)
            k[p + ".ff.net.1.weight"] = (DIM_HEAD_LQ, d)
            k[p + ".ff.net.1.bias"] = (DIM_HEAD_LQ,)
            k[p + ".ff.net.4.weight"] = (d, DIM_HEAD_LQ)
            k[p + ".ff.net.4.bias"] = (d,)
    for l in range(0 if cfg.learnable_query else cfg.fusion_layers):
        p = f"joints_late_fusion.attn_fusion.{l}"
        k[p + ".to_q.weight"] = (inner, d)
        k[p + ".to_k.weight"] = (inner, d)
        k[p + ".to_v.weight"] = (inner, d)
        k[p + ".to_out.weight"] = (d, inner)
        k[p + ".to_out.bias"] = (d,)
        k[p + ".norm1.weight"] = (d,)
        k[p + ".norm1.bias"] = (d,)
        k[p + ".norm2.weight"] = (d,)
        k[p + ".norm2.bias"] = (d,)
        k[p + ".ff.net.0.weight"] = (d,)
        k[p + ".ff.net.0.bias"] = (d,)
        k[p + ".ff.net.1.weight"] = (DIM_HEAD, d)
        k[p + ".ff.net.1.bias"] = (DIM_HEAD,)
        k[p + ".ff.net.4.weight"] = (d, DIM_HEAD)
        k[p + ".ff.net.4.bias"] = (d,)
    if cfg.use_gcn:
        for i, (ci, co) in enumerate([(d, 256), (256, 64), (64, 3)]):
            k[f"joints_decoder.joints_gcn{i + 1}.weight"] = (3, 1, ci, co)
            k[f"joints_decoder.joints_gcn{i + 1}.bias"] = (1, 1, co)
    else:
        k["joints_decoder.joints_fc1.weight"] = (64, d)
        k["joints_decoder.joints_fc1.bias"] = (64,)
        k["joints_decoder.joints_fc2.weight"] = (3, 64)
        k["joints_decoder.joints_fc2.bias"] = (3,)


def _hrnet_layout(k: "OrderedDict[str, tuple]", cfg: HotPathConfig) -> None:
    """HighResolutionNet module tree: backbones/hrnet.py:231-311 (ctor), 152-185 (fuse layers),
    287-311 (transition layers).  BatchNorm2d everywhere (never frozen)."""
    ch = HRNET_CHANNELS[cfg.backbone_type]
    k["backbone.conv1.weight"] = (64, 3, 3, 3)
    _bn(k, "backbone.bn1", 64, False)
    k["backbone.conv2.weight"] = (64, 64, 3, 3)
    _bn(k, "backbone.bn2", 64, False)
    inpl = 64
    for bi in range(4):   # stage 1: 4 Bottlenecks, planes 64
        p = f"backbone.layer1.{bi}"
        k[p + ".conv1.weight"] = (64, inpl, 1, 1)
        _bn(k, p + ".bn1", 64, False)
        k[p + ".conv2.weight"] = (64, 64, 3, 3)
        _bn(k, p + ".bn2", 64, False)
        k[p + ".conv3.weight"] = (256, 64, 1, 1)
        _bn(k, p + ".bn3", 256, False)
        if bi == 0:
            k[p + ".downsample.0.weight"] = (256, 64, 1, 1)
            _bn(k, p + ".downsample.1", 256, False)
        inpl = 256
    pre = [256]
    for st, (nmod, nbr) in enumerate([(1, 2), (4, 3), (3, 4)]):
        cur = ch[:nbr]
        tp = f"backbone.transition{st + 1}"
        for i in range(nbr):   # _make_transition_layer
            if i < len(pre):
                if cur[i] != pre[i]:
                    k[f"{tp}.{i}.0.weight"] = (cur[i], pre[i], 3, 3)
                    _bn(k, f"{tp}.{i}.1", cur[i], False)
            else:
                for j in range(i + 1 - len(pre)):
                    outc = cur[i] if j == i - len(pre) else pre[-1]
                    k[f"{tp}.{i}.{j}.0.weight"] = (outc, pre[-1], 3, 3)
                    _bn(k, f"{tp}.{i}.{j}.1", outc, False)
        for m in range(nmod):   # HighResolutionModule
            mp = f"backbone.stage{st + 2}.{m}"
            for b in range(nbr):
                for blk in range(4):
                    bp = f"{mp}.branches.{b}.{blk}"
                    k[bp + ".conv1.weight"] = (cur[b], cur[b], 3, 3)
                    _bn(k, bp + ".bn1", cur[b], False)
                    k[bp + ".conv2.weight"] = (cur[b], cur[b], 3, 3)
                    _bn(k, bp + ".bn2", cur[b], False)
            for i in range(nbr):
                for j in range(nbr):
                    fp = f"{mp}.fuse_layers.{i}.{j}"
                    if j > i:
                        k[fp + ".0.weight"] = (cur[i], cur[j], 1, 1)
                        _bn(k, fp + ".1", cur[i], False)
                    elif j < i:
                        for q in range(i - j):
                            outc = cur[i] if q == i - j - 1 else cur[j]
                            k[f"{fp}.{q}.0.weight"] = (outc, cur[j], 3, 3)
                            _bn(k, f"{fp}.{q}.1", outc, False)
        pre = cur


def executed_keys(cfg: HotPathConfig) -> List[str]:
    """Keys the forward actually reads (drops layer4/fc/num_batches_tracked)."""
    out = []
    for key in state_dict_layout(cfg):
        if key.endswith("num_batches_tracked"):
            continue
        if key.startswith("backbone.layer4") or key.startswith("backbone.fc"):
            continue
        out.append(key)
    return out


def remap_legacy_keys(state_dict: Dict[str, object]) -> Dict[str, object]:
    """Legacy checkpoint key remap, /root/reference/src/eval.py:15-52: a checkpoint is
    legacy iff it holds ``pose_net.conv.0.weight`` or ``sample_net.conv.0.weight``; then
    ``pose_net.conv.`` -> ``pose_net.`` and ``sample_net.`` -> ``sample_nets.0.``."""
    if not any(k in state_dict for k in ("pose_net.conv.0.weight", "sample_net.conv.0.weight")):
        return OrderedDict(state_dict)
    out = OrderedDict()
    for key, v in state_dict.items():
        out[key.replace("pose_net.conv.", "pose_net.").replace("sample_net.", "sample_nets.0.")] = v
    return out


def conv_flops_per_image(cfg: HotPathConfig, size: int | None = None) -> Dict[str, float]:
    """Dense algorithmic FLOPs (2*MAC) per image for the conv stack, by stage; matches
    SURVEY.md section 8(d) / Appendix A when size=256."""
    s = size or cfg.image_size
    out = {"stem": 2.0 * 64 * 3 * 49 * (s // 2) ** 2}
    blocks = RESNET_BLOCKS[cfg.backbone_type]
    inpl, h = 64, s // 4
    total = 0.0
    for li in range(3):
        planes = 64 << li
        stride = 1 if li == 0 else 2
        if cfg.is_paper and li == 2:
            stride = 1
        f = 0.0
        for bi in range(blocks[li]):
            st = stride if bi == 0 else 1
            ho = h // st
            if cfg.is_paper:
                f += 2.0 * planes * inpl * h * h
                f += 2.0 * planes * planes * 9 * ho * ho
                f += 2.0 * planes * 4 * planes * ho * ho
                outpl = planes * 4
            else:
                f += 2.0 * planes * inpl * 9 * ho * ho
                f += 2.0 * planes * planes * 9 * ho * ho
                outpl = planes
            if bi == 0 and (st != 1 or inpl != outpl):
                f += 2.0 * outpl * inpl * ho * ho
            inpl, h = outpl, ho
        out[f"layer{li + 1}"] = f
        total += f
    hm = h
    if cfg.is_paper:
        out["pose_net"] = 2.0 * (512 * cfg.backbone_channels[0] + 21 * 512) * hm * hm
        out["sample_net"] = 2.0 * (cfg.backbone_channels[0] // 2) * cfg.backbone_channels[0] * hm * hm
    else:
        h2 = 2 * hm
        out["pose_net"] = 2.0 * (128 * 256 * 16 * hm * hm + 64 * 128 * 9 * h2 * h2 + 21 * 64 * 9 * h2 * h2)
        f, hh = 0.0, hm
        for c in cfg.backbone_channels:
            f += 2.0 * (c // 2) * c * hh * hh
            hh *= 2
        out["sample_net"] = f
    return out
