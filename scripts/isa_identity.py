"""Are the device instruction streams of the convolution kernels the same as at another commit?  (How "the default build's device code is unchanged" is
checked when an experiment is added to a translation unit: the kernels that were measured on the GPU must compile to exactly the code that was measured.)

    python scripts/isa_identity.py <git-rev> [file.hip ...]          # default files: y7t_conv.hip y7t_conv_patch.hip

Compiles each file at <git-rev> (its csrc/ and include/ extracted to a temporary directory) and in the working tree to gfx950 assembly
(hipcc -S --cuda-device-only, the product's flags) and compares kernel by kernel: label numbers and the kernels' own mangled names are normalised (adding a
defaulted template parameter renames every instance), everything else must match.  Kernels that exist only on one side are listed, not counted as differences.
Exit code 1 if an instance that exists on both sides differs."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "-S", "--cuda-device-only"]      # (the conv translation units' flags, yolov7-tracker_amd/build.py)


def asm(csrc, include, f, out):
    subprocess.run([HIPCC] + FLAGS + ["-I", include, "-I", csrc, os.path.join(csrc, f), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out


def kernels(path):
    t = open(path).read().splitlines()
    out, i = {}, 0
    while i < len(t):
        m = re.match(r"^(_Z\S+):", t[i])
        if m and "@" in t[i]:
            name, j = m.group(1), i + 1
            while j < len(t) and not t[j].startswith(".Lfunc_end"):
                j += 1
            body = [re.sub(r"\.LBB\d+_", ".LBB_", l.split(";")[0].rstrip()) for l in t[i + 1:j]]
            body = [re.sub(r"_Z[0-9A-Za-z_]*k_conv\w+", "KERNEL", l) for l in body if l.strip() and not l.strip().startswith((".loc", ".file", ".cfi"))]
            out[name] = body
            i = j
        i += 1
    return out


def main():
    rev = sys.argv[1]
    files = sys.argv[2:] or ["y7t_conv.hip", "y7t_conv_patch.hip"]
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        old_csrc, old_inc = os.path.join(tmp, "yolov7-tracker_amd", "csrc"), os.path.join(tmp, "include")      # (csrc includes "../../include/y7t.h")
        os.makedirs(old_csrc), os.makedirs(old_inc)
        for d, dst in (("yolov7-tracker_amd/csrc", old_csrc), ("include", old_inc)):
            names = subprocess.check_output(["git", "ls-tree", "--name-only", rev, d + "/"], cwd=ROOT, text=True).split()
            for n in names:
                open(os.path.join(dst, os.path.basename(n)), "wb").write(subprocess.check_output(["git", "show", "%s:%s" % (rev, n)], cwd=ROOT))
        for f in files:
            a = kernels(asm(old_csrc, old_inc, f, os.path.join(tmp, "old.s")))
            b = kernels(asm(os.path.join(ROOT, "yolov7-tracker_amd", "csrc"), os.path.join(ROOT, "include"), f, os.path.join(tmp, "new.s")))
            same = diff = 0
            gone = []
            for n, body in a.items():
                cand = [n] + [n.replace("EEv11Y7TConvArgs", "ELi%dEEv11Y7TConvArgs" % d) for d in (4, 0)]      # a defaulted trailing int parameter added since (NW = 4)
                hit = next((c for c in cand if c in b), None)
                if hit is None:
                    gone.append(n)
                elif b[hit] == body:
                    same += 1
                else:
                    diff += 1
                    print("DIFFERENT", f, n)
            new = len(b) - same - diff
            print("%s vs %s: %d kernels identical, %d different, %d only at %s, %d only in the working tree" % (f, rev, same, diff, len(gone), rev, new))
            bad += diff
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
