#!/usr/bin/env python3
"""Dev tool: build A/B variants of libisdf_hip.so (compile-time switches of the chain / dW kernels) into
variants/lib_<name>.so; tools/ab_bench.sh runs them back to back on the SAME GPU box.
usage: python tools/build_variants.py name1="-DFLAG=1 ..." name2="..."     (base = no flags is always built)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isdf_amd import build as b

def main():
    b.build(force=False, verbose=False)
    os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
    variants = [("base", "")] + [tuple(a.split("=", 1)) for a in sys.argv[1:]]
    objs = {s: os.path.join(b.HERE, "build", s.replace(".hip", ".o")) for s in b.SOURCES}
    procs = []
    for name, flags in variants:
        mine = dict(objs)
        jobs = []
        for src in ("chain.hip", "dw.hip", "optim.hip", "sampler.hip", "ingest.hip", "capi.hip"):
            if not flags:
                continue
            o = os.path.join(ROOT, "variants", "%s_%s.o" % (name, src[:-4]))
            mine[src] = o
            jobs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + b.PER_FILE.get(src, []) + flags.split() + ["-c", os.path.join(b.CSRC, src), "-o", o]))
        procs.append((name, mine, jobs))
    for name, mine, jobs in procs:
        for j in jobs:
            assert j.wait() == 0, name
        lib = os.path.join(ROOT, "variants", "lib_%s.so" % name)
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [mine[s] for s in b.SOURCES])
        print("built", lib)

if __name__ == "__main__":
    main()
