"""Phase timeline of csrc/gemm1x1_mfma.hip from in-kernel stamps (a -DG1_STAMP build: s_memrealtime at 100 MHz at the phase
boundaries of every workgroup, written to a buffer of its own).  Build and run on the GPU box:

    cd rs_detection_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DG1_STAMP -I. \
        -c gemm1x1_mfma.hip -o /tmp/g1_stamp.o && hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/librsdet_g1stamp.so \
        $(ls *.o | grep -v gemm1x1_mfma.o) /tmp/g1_stamp.o
    RSDET_LIB_PATH=/tmp/librsdet_g1stamp.so python3 profiles/scripts/gemm1x1_stamps.py

Per shape: medians over workgroups of (a) start -> prologue issued, (b) per step: wait for the step's data (from the end of
the previous phase), MFMA phase, (c) per tile end: wait for the identity operand, epilogue + stores issue; all in us."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd import _lib as _L  # noqa: E402

lib = ctypes.CDLL(_L.LIB_PATH)
dev = torch.device("cuda:0")
shapes = [(4 * 128 * 128, 512, 128, True), (4 * 128 * 128, 128, 512, False), (4 * 256 * 256, 256, 64, True),
          (4 * 64 * 64, 1024, 256, True)]
for (M, N, K, res) in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16() if res else None
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    st = [torch.rand(N, device=dev) + 0.5 for _ in range(4)]
    buf = (ctypes.c_ulonglong * (512 * 64))()

    def run():
        rc = _L.load().rsdet_conv1x1_bn_act_fwd_bf16(_L.ptr(x), _L.ptr(w), M, N, K, _L.ptr(st[0]), _L.ptr(st[1]), _L.ptr(st[2]),
                                                     _L.ptr(st[3]), 1e-5, _L.ptr(r), 1, _L.ptr(y), _L.stream_ptr())
        assert rc == 0
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    lib.rsdet_g1_stamps_read(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.rsdet_g1_stamps_read(buf, 0)
    t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 64).astype(np.float64)
    live = t[:, 0] > 0
    t = t[live]
    base = t[:, 0].min()
    us = (t - base) / 100.0                                  # 100 MHz -> us
    us[t == 0] = np.nan
    KS = K // 64
    print("== M %d N %d K %d res %d: %d workgroups, launch %.1f us (events), last workgroup ends at %.1f us" % (
        M, N, K, res, len(t), e0.elapsed_time(e1) * 1e3, np.nanmax(us[:, 63])))
    print("   start spread %.1f us; start -> prologue issued %.2f us (median)" % (
        np.nanmax(us[:, 0]), np.nanmedian(us[:, 1] - us[:, 0])))
    prev = us[:, 1]
    for s in range(15):
        a, b = us[:, 2 + 4 * s], us[:, 3 + 4 * s]
        if np.all(np.isnan(a)):
            break
        line = "   step %2d: wait %.2f  mfma %.2f" % (s, np.nanmedian(a - prev), np.nanmedian(b - a))
        prev = b
        if (s % KS) == KS - 1:
            c, d = us[:, 4 + 4 * s], us[:, 5 + 4 * s]
            if res:
                line += "  | tile end: wait identity %.2f  epilogue+stores %.2f" % (np.nanmedian(c - b), np.nanmedian(d - c))
                prev = d
            else:
                line += "  | tile end: epilogue+stores %.2f" % np.nanmedian(d - b)
                prev = d
        print(line)
    print("   end of loop -> kernel end %.2f us" % np.nanmedian(us[:, 63] - prev))
