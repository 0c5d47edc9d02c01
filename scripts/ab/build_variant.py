"""exp_libs/<tag>.so = the current library with ONE source file replaced (kernel A/B on one box: AADG_LIB_PATH=exp_libs/<tag>.so).
    python scripts/ab/build_variant.py <tag> <file.hip> [git-rev]     (git-rev: take the file from that revision; default: working tree)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import build as b

tag, rel = sys.argv[1], sys.argv[2]
rev = sys.argv[3] if len(sys.argv) > 3 else None
b.build_hip()
os.makedirs(os.path.join(ROOT, "exp_libs"), exist_ok=True)
src = os.path.join(ROOT, rel)
if rev:
    text = subprocess.check_output(["git", "-C", ROOT, "show", "%s:%s" % (rev, rel)])
    src = os.path.join(os.path.dirname(src), "_variant_" + os.path.basename(rel))
    open(src, "wb").write(text)
obj = "/tmp/_variant_%s.o" % tag
subprocess.check_call([b._hipcc()] + b.HIPCC_FLAGS + ["-c", src, "-o", obj])
if rev:
    os.remove(src)
objdir = os.path.join(b.LIB_DIR, "obj")
objs = [obj if o == os.path.basename(rel)[:-4] + ".o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir)) if o.endswith(".o")]
out = os.path.join(ROOT, "exp_libs", tag + ".so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
