#!/usr/bin/env python3
"""List every packed-fp32 instruction with an op_sel modifier in the built objects, per kernel (the guard test's view)."""
import os, re, shutil, subprocess, sys, tempfile
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "haconvdr_amd", "csrc")
def scan(obj):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.join(CSRC, obj), tmp)
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=tmp, check=True, capture_output=True)
        code = [f for f in os.listdir(tmp) if f.startswith(obj + ".") and "gfx950" in f]
        dis = subprocess.run([OBJDUMP, "-d", code[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    name = None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            name = m.group(1)
        elif name and re.search(r"v_pk_(mul|fma|add)_f32", line) and "op_sel:[" in line:
            out.setdefault(name, []).append(line.strip().split("//")[0].strip())
    return out
if __name__ == "__main__":
    for obj in ("encoder.o", "flat_ip.o"):
        for k, v in scan(obj).items():
            print(obj, len(v), k)
            if "-v" in sys.argv:
                for e in v[:6]:
                    print("    ", e)
