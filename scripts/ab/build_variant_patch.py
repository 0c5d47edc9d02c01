"""exp_libs/<tag>.so from a PATCHED copy of one source file: python scripts/ab/build_variant_patch.py <tag> <file.hip> <old> <new> [<old> <new> ...]
(each <old> must occur exactly once).  For timing experiments that switch a part of a kernel off; the tree is not touched."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import build as b
tag, rel = sys.argv[1], sys.argv[2]
text = open(os.path.join(ROOT, rel)).read()
pairs = sys.argv[3:]
for old, new in zip(pairs[0::2], pairs[1::2]):
    assert text.count(old) == 1, (old, text.count(old))
    text = text.replace(old, new)
b.build_hip()
src = os.path.join(ROOT, os.path.dirname(rel), "_variant_" + os.path.basename(rel))
open(src, "w").write(text)
obj = "/tmp/_variant_%s.o" % tag
try:
    subprocess.check_call([b._hipcc()] + b.HIPCC_FLAGS + ["-c", src, "-o", obj])
finally:
    os.remove(src)
objdir = os.path.join(b.LIB_DIR, "obj")
objs = [obj if o == os.path.basename(rel)[:-4] + ".o" else os.path.join(objdir, o) for o in sorted(os.listdir(objdir)) if o.endswith(".o")]
os.makedirs(os.path.join(ROOT, "exp_libs"), exist_ok=True)
out = os.path.join(ROOT, "exp_libs", tag + ".so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
