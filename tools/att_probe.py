"""Launch time of the attention kernel alone (hmv_op_attention): cfg-3's fusion shapes -- B = 32 samples, 8 views x 21 joints = 168
tokens (self-attention, then 21 queries against 168 keys, then 21 x 21).  `HMV_LIB=<other build>` for a same-box A/B."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from handmvnet_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    op = lib.hmv_op_attention_x3 if "--x3" in sys.argv else lib.hmv_op_attention   # --x3: the fp16-kernel modes' (hi, lo) form
    dev = torch.device("cuda:0")
    for (B, T, Tq, koff, Tk) in ((32, 168, 168, 0, 168), (32, 168, 21, 0, 168), (32, 21, 21, 0, 21), (1, 168, 168, 0, 168), (32, 273, 273, 0, 273)):
        g = torch.Generator(device="cpu").manual_seed(1)
        qkv = torch.randn(B, T, 3072, generator=g).to(dev)
        out = torch.empty(B, Tq, 1024, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            rc = op(0, ctypes.c_void_p(qkv.data_ptr()), B, T, Tq, koff, Tk, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))
            assert rc == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            op(0, ctypes.c_void_p(qkv.data_ptr()), B, T, Tq, koff, Tk, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(s))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        flops = 4.0 * B * 8 * Tq * Tk * 128
        print(f"B={B} T={T} Tq={Tq} Tk={Tk}: {us:7.1f} us  {flops / us * 1e-6:6.1f} TFLOP/s  checksum {out.float().sum().item():.6e}")


if __name__ == "__main__":
    main()
