#!/usr/bin/env python3
"""Build-time check of the row-group kernels' ISA (called by spasm_amd/csrc/Makefile on every .so it links).

schur_group_kernel keeps its prefetch ring in accumulation registers (AGPRs) written by inline-asm loads the
compiler does not know about.  That is only sound while the compiler never touches AGPRs itself in these
kernels: no v_accvgpr_write, no AGPR above the ring, no scratch (a spill could be parked in an AGPR), and the only
instructions naming an AGPR are the ring's own loads and v_accvgpr_read.  A compiler upgrade, a flag change or extra
register pressure would break that silently, so the build fails here instead.

usage: check_isa.py <shared object>   (exit status 0 = every variant is clean)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def check(so_path, expect_variants=12):
    if not os.path.exists(OBJDUMP):
        raise SystemExit("check_isa: %s not found (cannot verify the AGPR ring; refusing to accept the build)" % OBJDUMP)
    problems = []
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, so)
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        kernels = {}
        for f in sorted(os.listdir(tmp)):
            if not f.endswith("gfx950"):
                continue
            text = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if "schur_group_kernel" in m.group(1) else None
                    if cur:
                        kernels[cur] = []
                elif cur and line.strip():
                    kernels[cur].append(line)
    if len(kernels) < expect_variants:
        problems.append("only %d schur_group_kernel variants found (expected %d)" % (len(kernels), expect_variants))
    for name, lines in kernels.items():
        wide = "ILb1E" in name.split("schur_group_kernel")[1][:6]
        limit = 64 if wide else 48
        for line in lines:
            ins = line.split("//")[0]
            if "v_accvgpr_write" in ins or "scratch_" in ins:
                problems.append("%s: %s" % (name, ins.strip()))
                continue
            regs = [int(x) for x in re.findall(r"\ba\[?(\d+)", ins)]
            if regs and (max(regs) >= limit or not re.search(r"global_load_dword|v_accvgpr_read", ins)):
                problems.append("%s: %s" % (name, ins.strip()))
    return len(kernels), problems


if __name__ == "__main__":
    n, problems = check(sys.argv[1])
    if problems:
        print("check_isa: %s: the compiler touched the AGPR prefetch ring of the row-group kernels:" % sys.argv[1])
        for p in problems[:40]:
            print("   ", p)
        sys.exit(1)
    print("check_isa: %s: %d row-group kernel variants, AGPR ring untouched by the compiler" % (sys.argv[1], n))
