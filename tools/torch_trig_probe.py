#!/usr/bin/env python
"""What PyTorch-CPU's float32 sin / cos / tan / atan / atan2 are, measured (BUILD CONTAINER ONLY: needs torch on the CPU and gcc).

Compares torch's results with (a) the correctly rounded value (float)libm(double) and (b) the SLEEF functions libtorch_cpu.so itself
exports (Sleef_{sin,cos,tan,atan,atan2}f16_u10 / _u35, called through a small C shim compiled on the fly -- AVX-512 hosts only).
Result in this container (torch 2.10.0+rocm7.0, MKL 2024.2, AVX-512; recorded in include/sigma_trig_f32.h and DESIGN.md section 2):
  sin / cos / tan / atan: within 1 ulp of the correctly rounded value (4.9 % / 5.0 % / 0.6 % / 0.05 % of the results differ from it), equal to
  NEITHER SLEEF function (u10: 1.9 % / 2.2 % / 12 % / 1.8 % differ) -> a closed vector math library, not restatable;
  atan2: equal to Sleef_atan2f16_u10 in the vectorised body of a tensor, to glibc's atan2f (~ correctly rounded) in its scalar tail.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import torch

SHIM = r'''
#include <immintrin.h>
#define D(n) __m512 Sleef_##n##f16_u10(__m512); __m512 Sleef_##n##f16_u35(__m512);
D(sin) D(cos) D(tan) D(atan)
__m512 Sleef_atan2f16_u10(__m512, __m512);
#define A(n, v) void n##_##v(const float* x, float* y, long c) { for (long i = 0; i + 16 <= c; i += 16) _mm512_storeu_ps(y + i, Sleef_##n##f16_##v(_mm512_loadu_ps(x + i))); }
A(sin, u10) A(cos, u10) A(tan, u10) A(atan, u10) A(sin, u35) A(cos, u35) A(tan, u35) A(atan, u35)
void atan2_u10(const float* a, const float* b, float* y, long c) { for (long i = 0; i + 16 <= c; i += 16) _mm512_storeu_ps(y + i, Sleef_atan2f16_u10(_mm512_loadu_ps(a + i), _mm512_loadu_ps(b + i))); }
'''


def main():
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    d = tempfile.mkdtemp()
    src, so = os.path.join(d, "shim.c"), os.path.join(d, "shim.so")
    open(src, "w").write(SHIM)
    lib = None
    try:
        subprocess.check_call(["gcc", "-O2", "-mavx512f", "-fPIC", "-shared", "-o", so, src, "-L" + tl, "-ltorch_cpu", "-Wl,-rpath," + tl])
        lib = C.CDLL(so)
    except Exception as exc:  # noqa: BLE001
        print("SLEEF shim unavailable:", exc)
    rng = np.random.default_rng(1)
    n = 1 << 20
    x = rng.uniform(-7, 7, n).astype(np.float32)
    print(torch.__version__, "| CPU capability:", torch.backends.cpu.get_cpu_capability())
    for nm, fn, npf in (("sin", torch.sin, np.sin), ("cos", torch.cos, np.cos), ("tan", torch.tan, np.tan), ("atan", torch.atan, np.arctan)):
        t = fn(torch.from_numpy(x)).numpy()
        cr = npf(x.astype(np.float64)).astype(np.float32)
        row = f"{nm:5s} torch != correctly rounded: {(t != cr).mean():.4%} (max {np.abs(t.view(np.int32).astype(np.int64) - cr.view(np.int32)).max()} ulp)"
        if lib is not None:
            for v in ("u10", "u35"):
                y = np.zeros_like(x)
                getattr(lib, f"{nm}_{v}")(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(n))
                row += f" | torch != SLEEF {v}: {(t.view(np.uint32) != y.view(np.uint32)).mean():.4%}"
        print(row)
    a, b = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    t = torch.atan2(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    cr = np.arctan2(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    row = f"atan2 torch != correctly rounded: {(t != cr).mean():.4%}"
    if lib is not None:
        y = np.zeros_like(a)
        lib.atan2_u10(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(n))
        row += f" | torch != SLEEF u10 (vector body): {(t.view(np.uint32) != y.view(np.uint32)).mean():.4%}"
    tail = torch.atan2(torch.tensor([3.0, 1.0, 0.1]), torch.tensor([3.0, 1.0, 0.1])).numpy()
    row += f" | 3-element tensor atan2(v, v) == float32(pi / 4): {bool((tail == np.float32(np.pi / 4)).all())} (scalar tail, not SLEEF)"
    print(row)


if __name__ == "__main__":
    main()
