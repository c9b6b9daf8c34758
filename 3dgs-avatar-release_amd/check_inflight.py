"""Build-time check of render_bwd.hip's entry switch (split_accumulate / switch_entry): exec-masked LDS reads land straight in
the working registers of the lanes that switch entries, while other instructions run in their shadow.  Since round 4 every
such read is issued AND waited for inside one inline-asm statement, so the compiler cannot put anything in between; this
check stays as the gate that says so about the code that was actually generated: it disassembles the gfx950 code object of
render_bwd.o and fails if any instruction between a ds_read_b128 group and the following `s_waitcnt lgkmcnt(0)` names one of
the destination registers.  Run by build.py after every compile (it lives in the package, next to build.py, and finds
llvm-objdump next to the hipcc that compiled the object) and by tests/test_cabi_host.py on the built object."""
import os
import re
import subprocess
import sys
import tempfile

def _objdump(hipcc=None):
    """llvm-objdump of the toolchain that built the object: next to the resolved hipcc (<rocm>/bin/hipcc ->
    <rocm>/lib/llvm/bin), else on PATH, else the default ROCm location."""
    import shutil
    cands = []
    exe = shutil.which(hipcc) if hipcc else None
    if exe:
        root = os.path.dirname(os.path.dirname(os.path.realpath(exe)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    cands += [shutil.which("llvm-objdump") or "", "/opt/rocm/lib/llvm/bin/llvm-objdump"]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("llvm-objdump not found (looked next to %r, on PATH and under /opt/rocm)" % (hipcc,))


def check(obj, hipcc=None):
    objdump = _objdump(hipcc)
    tmp = tempfile.mkdtemp(prefix="gs_inflight_")
    local = os.path.join(tmp, "render_bwd.o")
    with open(obj, "rb") as f, open(local, "wb") as g:
        g.write(f.read())
    subprocess.run([objdump, "--offloading", local], cwd=tmp, capture_output=True, check=True)
    dev = [f for f in os.listdir(tmp) if "hipv4-amdgcn" in f]
    if not dev:
        raise RuntimeError("no gfx950 code object found in %s" % obj)
    asm = subprocess.run([objdump, "-d", os.path.join(tmp, dev[0])], capture_output=True, text=True, check=True).stdout.split("\n")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
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
