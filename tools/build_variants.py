#!/usr/bin/env python3
"""Dev tool: build A/B variants of libisdf_hip.so into variants/lib_<name>.so; tools/ab_bench.sh runs them back to back on
the SAME GPU box.  A variant is either a set of extra compiler flags or a PATCH of the kernel sources (the shipped sources
carry no `#if` experiment switches):

    python tools/build_variants.py name1="-DFLAG=1 ..." name2=@tools/variants/some.patch      (base = the tree as it is, always built)

A patch is applied (`patch -p1`, paths as `git diff` writes them) to a scratch copy of isdf_amd/csrc + include/.
A spec of the form  name=sed:FILE:SCRIPT  runs `sed -E SCRIPT` over csrc/FILE (FILE = * : every file) of the scratch copy instead (one-line variants that
survive edits of the surrounding code; ISDF_VARIANT_EXTRA_FLAGS adds compiler flags to every variant of the call, e.g. -DISDF_DEBUG_HOOKS=1
for a source variant with stage stamps), e.g. the A/B partner of the pair-tile forward kernel:
    onetile='sed:fwd_pair.hip:s/^(bool fwd_pair_supported\(const NetLayout& l\) \{).*$/\1 (void)l; return false; }/'"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isdf_amd import build as b


def patched_sources(patch):
    tmp = tempfile.mkdtemp(prefix="isdf_variant_")
    os.makedirs(os.path.join(tmp, "isdf_amd"))
    shutil.copytree(b.CSRC, os.path.join(tmp, "isdf_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    if patch.startswith("sed:"):
        _, fname, script = patch.split(":", 2)
        cdir = os.path.join(tmp, "isdf_amd", "csrc")
        targets = [os.path.join(cdir, f) for f in sorted(os.listdir(cdir))] if fname == "*" else [os.path.join(cdir, fname)]
        before = [open(t).read() for t in targets]
        for t in targets:
            subprocess.check_call(["sed", "-E", "-i", script, t])
        assert [open(t).read() for t in targets] != before, "sed variant changed nothing: " + patch
    else:
        subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.abspath(patch)], cwd=tmp)
    return os.path.join(tmp, "isdf_amd", "csrc")


def main():
    b.build(force=False, verbose=False)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    variants = [("base", "")] + [tuple(a.split("=", 1)) for a in sys.argv[1:]]
    objs = {s: os.path.join(b.HERE, "build", s.replace(".hip", ".o")) for s in b.SOURCES}
    procs = []
    for name, spec in variants:
        mine = dict(objs)
        jobs = []
        csrc, flags = b.CSRC, spec
        if spec.startswith("@"):
            csrc, flags = patched_sources(spec[1:]), ""
        elif spec.startswith("sed:"):
            csrc, flags = patched_sources(spec), ""
        for src in b.SOURCES:
            if not spec:
                continue
            o = os.path.join(ROOT, "variants", "%s_%s.o" % (name, src[:-4]))
            mine[src] = o
            jobs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + b.PER_FILE.get(src, []) + flags.split() + os.environ.get("ISDF_VARIANT_EXTRA_FLAGS", "").split()
                                         + ["-c", os.path.join(csrc, src), "-o", o]))
        procs.append((name, mine, jobs))
    for name, mine, jobs in procs:
        for j in jobs:
            assert j.wait() == 0, name
        lib = os.path.join(ROOT, "variants", "lib_%s.so" % name)
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [mine[s] for s in b.SOURCES])
        print("built", lib)


if __name__ == "__main__":
    main()
