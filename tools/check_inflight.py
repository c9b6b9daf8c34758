"""Build-time check of render_bwd.hip's entry switch (split_accumulate): its LDS reads are issued in one inline-asm block
and waited for at the end of the next one, and between the two the compiler must not read, write or copy the destination
registers (it does not know that loads are in flight there).  Disassembles the gfx950 code object of build/render_bwd.o and
fails if any instruction between a ds_read_b128 triple and the following `s_waitcnt lgkmcnt(0)` names one of those
registers.  Run by build.py after every compile of render_bwd.hip."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def check(obj):
    tmp = tempfile.mkdtemp(prefix="gs_inflight_")
    local = os.path.join(tmp, "render_bwd.o")
    with open(obj, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, capture_output=True, check=True)
    dev = [f for f in os.listdir(tmp) if "hipv4-amdgcn" in f]
    if not dev:
        raise RuntimeError("no gfx950 code object found in %s" % obj)
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, dev[0])], capture_output=True, text=True,
                         check=True).stdout.split("\n")
    windows = bad = 0
    i = 0
    while i < len(asm):
        if "ds_read_b128" in asm[i]:
            j = i
            while j < len(asm) and "lgkmcnt(0)" not in asm[j]:
                j += 1
            regs = set()
            k = i
            while k < j and "ds_read_b" in asm[k]:  # the two or three reads of one switch (b128, b128, b128 | b32)
                m = re.search(r"ds_read_b128 v\[(\d+):(\d+)\]", asm[k])
                if m:
                    regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                else:
                    regs.add(int(re.search(r"ds_read_b\d+ v(\d+)", asm[k]).group(1)))
                k += 1
            for line in asm[k:j]:
                ins = line.split("//")[0]
                used = {int(r) for r in re.findall(r"\bv(\d+)\b", ins)}
                for m in re.finditer(r"v\[(\d+):(\d+)\]", ins):
                    used |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                if used & regs:
                    bad += 1
                    print("in-flight register touched: %s" % line.strip(), file=sys.stderr)
            windows += 1
            i = j
        i += 1
    if windows == 0:
        raise RuntimeError("no LDS read windows found: has the entry switch of render_bwd.hip changed?")
    if bad:
        raise RuntimeError("%d instruction(s) touch registers whose LDS loads are in flight (render_bwd.hip, split_accumulate)" % bad)
    return windows


if __name__ == "__main__":
    print("%d windows checked" % check(sys.argv[1]))
