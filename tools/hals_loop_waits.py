"""Memory instructions and waits inside the unrolled block loop of the H row sweep, from a compiled ISA listing:
python tools/hals_loop_waits.py file.s  -- a wait inside the steps (other than the mid-block drain) stalls every block"""
import re, sys
txt = open(sys.argv[1]).read()
for kern in ("hals_h_persist_kernel", "hals_h_stage_kernel"):
    m = re.search(r"^_Z\d+%s.*?\.end_amdhsa_kernel" % kern, txt, re.S | re.M)
    if not m:
        continue
    L = [l.rstrip() for l in m.group(0).split("\n")]
    idx = [i for i, l in enumerate(L) if "v_writelane_b32" in l]
    first, last = idx[0], idx[63]
    k = first
    while "Loop Header" not in L[k] and k > 0:
        k -= 1
    print(kern)
    for i in range(k, last + 40):
        l = L[i]
        if any(x in l for x in ("s_waitcnt", "global_", "s_sleep")) and not l.strip().startswith(";"):
            print(f"   {i - k:5d} step {sum(1 for j in idx[:64] if j <= i):2d}  {l.strip()}")
