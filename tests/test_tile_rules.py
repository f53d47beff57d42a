"""The launcher's tile rules (conv_igemm.hip::conv_pick_tile) for the shapes they were measured on: host logic, no GPU.  Every
expectation below is a measured choice recorded in DESIGN.md (forced-tile runs of bench.py --per-layer); a change here should come
with a new measurement."""
import pytest

from handmvnet_amd import _lib


def rule(M, Cout, K, f16=False, res=False):
    return _lib.load().hmv_tile_rule(M, Cout, K, int(f16), int(res)).decode()


FRAMES = 256                      # BASELINE configs[2]: 32 samples x 8 views
PX8 = FRAMES * 32 * 32            # pixels of the H/8 maps
PX4 = FRAMES * 64 * 64


@pytest.mark.parametrize("args,want", [
    # ---- ResNet50-paper, fp32, cfg-3
    ((PX8, 256, 2304), "256x256"),                       # layer3 conv2 (the dominant family)
    ((PX8, 256, 1024), "256x256"),                       # layer3 conv1
    ((PX8, 1024, 256, False, True), "256x128,k16,w8"),   # layer3 conv3 if conv_stream_f32 is switched off
    ((PX8, 128, 512), "256x128,k16,w8"),                 # layer2 conv1: the paired short tiles (round 3)
    ((PX4, 128, 256), "256x128,k16,w8"),                 # layer2.0 conv1
    ((PX4, 256, 64, False, True), "128x128,k16"),        # layer1 conv3 if conv_stream_f32 is switched off
    # ---- one 8-view sample (batch 1)
    ((8 * 1024, 256, 2304), "128x64"),                   # layer3 conv2: every CU gets a 128 x 64 tile
    ((8 * 1024, 1024, 256, False, True), "64x64"),       # layer3 conv3: little work per big tile
    ((8 * 4096, 256, 64, False, True), "64x64"),         # layer1 conv3
    ((5376 // 32, 524, 1024), "64x64"),                  # a token GEMM
    # ---- HRNet-w40, fp32: channel counts that are multiples of no wide tile
    ((FRAMES * 256, 160, 1440), "128x32"),               # 160-channel 3x3
    ((FRAMES * 1024, 80, 360), "128x32"),                # a 3x3 s2 fuse layer 40 -> 80
    ((FRAMES * 1024, 80, 2304), "256x128"),              # the long reduction 3x3 256 -> 80 keeps its tile
    ((FRAMES * 64, 320, 2880), "64x64"),                 # 320 channels over 16 384 pixels: five 64 x 64 tiles per CU
    # ---- fp16: the picks of round 2 stand (re-measured)
    ((PX8, 256, 2304, True), "256x256"),
    ((FRAMES * 256, 160, 1440, True), "256x256"),        # (launch_conv turns it into the 256 x 192 dense tile)
    ((FRAMES * 64, 320, 2880, True), "128x64"),
])
def test_tile_rules(args, want):
    assert rule(*args) == want, (args, rule(*args))
