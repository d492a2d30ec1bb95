"""Per-kernel instruction statistics of a gfx950 assembly dump (`hipcc -save-temps`, tools/build_dev.sh): total
instructions, scratch traffic, AGPR moves, LDS accesses, FP64 arithmetic, selects - for the kernels whose mangled name
contains the given substring.  usage: python tools/isa_kernel_stats.py /tmp/isa/qc_balance-hip-amdgcn-amd-amdhsa-gfx950.s EqpDenseE"""
import collections
import re
import sys

L = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(L) if re.match(r"^_ZN2qc\d+balance\w*:", l)]
for i, name in starts:
    if pat not in name:
        continue
    j = next(k for k in range(i, len(L)) if L[k].startswith(".Lfunc_end"))
    lines = [l.strip() for l in L[i + 1:j] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(l.split()[0] for l in lines)
    key = lambda *ps: sum(v for k, v in c.items() if k.startswith(ps))  # noqa: E731
    print(name[6:80], "| total", len(lines), "scratch", key("scratch_"), "accvgpr", key("v_accvgpr"), "ds_read", key("ds_read"), "ds_write", key("ds_write"),
          "f64", key("v_fma", "v_mul_f64", "v_add_f64"), "cndmask", key("v_cndmask"), "rcp/rsq", key("v_rcp", "v_rsq"), "mov", key("v_mov"))
