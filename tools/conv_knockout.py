"""Where does the conv kernels' time go?  Full-chip knock-out timings (wrong results, timing only).

  python tools/conv_knockout.py build      # here: compiles tools/ko/libcmf_ko{0,1,2,3}.so  (-DCMF_CONV_KNOCKOUT=k)
  python tools/conv_knockout.py run [T]    # on the GPU box: times the conv modes with each of them

k = 0: the product kernels; 1: W operand loaded for lag 0 only (no W stream in the lag loop); 2: one H operand read
from LDS per lag instead of sixteen; 3: both (the lag loop is MFMAs only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KO = os.path.join(ROOT, "tools", "ko")


def build():
    from importlib import import_module
    b = import_module("cmf_jl_amd.build")
    os.makedirs(KO, exist_ok=True)
    for k in range(4):
        out = os.path.join(KO, f"libcmf_ko{k}.so")
        cmd = [b.hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
               "-I", os.path.join(ROOT, "include"), "-I", b.CSRC] + ([f"-DCMF_CONV_KNOCKOUT={k}"] if k else []) + b.SOURCES + ["-o", out]
        subprocess.check_call(cmd)
        print(out, flush=True)


def run_one(k, T):
    import cmf_jl_amd._lib as L
    L.LIB_PATH = os.path.join(KO, f"libcmf_ko{k}.so")
    import cmf_jl_amd as cmf
    data = cmf.gen_synthetic(N=2000, T=T, seed=1234)
    W0, H0 = cmf.init_rand(data, L=20, K=32, seed=0)
    rule = cmf.MultUpdate(data, W0, H0)
    for variant in (2, 3):
        rule.set_option("conv_kernel", variant)
        out = []
        for name in ("conv", "conv_t", "conv_loss", "conv_loss_store"):
            ms = sorted(rule.time_kernel(name, 20)[0] for _ in range(5))
            out.append(f"{name} {ms[0]:.4f}")
        print(f"knockout={k} conv_kernel={variant}: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "one":
        run_one(int(sys.argv[2]), int(sys.argv[3]))
    else:
        T = sys.argv[2] if len(sys.argv) > 2 else "50000"
        for k in range(4):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", str(k), T], timeout=300)
