#!/usr/bin/env python3
"""SIMD issue-port model of the tile kernels from a tools/pmc_issue_mix.sh summary (rocprofv3 --pmc, separate passes).

    python tools/issue_model.py gpurun_out/pmc_issue/summary.csv  >  profiles/rNN_issue_port.json

Measured on MI355X by tools/probes/valu_rate.hip (profiles/r05_probe_valu_rate.txt): in an instruction stream that contains MFMAs,
every plain VALU wave-instruction costs ~4 issue cycles of its SIMD, a transcendental ~8, an MFMA ~8 of the 32 it executes for --
whatever the number of waves on the SIMD; the matrix pipe itself runs beside that.  A kernel's time on a SIMD is therefore bounded
below by BOTH  32 x N_mfma  (matrix pipe)  and  issue = valu_cycles + 8 N_mfma + 4 (N_lds + N_vmem)  (issue port), where
valu_cycles = 4 x SQ_ACTIVE_INST_VALU (the counter is in quad-cycles and already weighs transcendentals double).  The elapsed
SIMD-cycles of the dispatch are GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs."""
import csv, json, sys, collections

def main(path):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d[r["kernel"]][r["counter"]] = float(r["avg_per_dispatch"])
    out = {"source": path, "model": "issue = 4*SQ_ACTIVE_INST_VALU + 8*SQ_INSTS_MFMA + 4*(SQ_INSTS_LDS + SQ_INSTS_VMEM_RD + SQ_INSTS_VMEM_WR) "
                                     "SIMD-cycles; elapsed = GRBM_GUI_ACTIVE/8 * 1024; matrix = 32*SQ_INSTS_MFMA (profiles/r05_probe_valu_rate.txt)"}
    for k, c in d.items():
        need = ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU")
        if any(n not in c for n in need):
            continue
        elapsed = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        valu = 4.0 * c["SQ_ACTIVE_INST_VALU"]
        issue = valu + 8.0 * c["SQ_INSTS_MFMA"] + 4.0 * (c["SQ_INSTS_LDS"] + c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"])
        out[k] = {"elapsed_simd_cycles": elapsed, "issue_cycles": issue, "issue_port_utilisation": round(issue / elapsed, 4),
                  "valu_issue_cycles": valu, "valu_cycles_per_instruction": round(valu / c["SQ_INSTS_VALU"], 3),
                  "matrix_pipe_cycles": 32.0 * c["SQ_INSTS_MFMA"], "matrix_pipe_utilisation": round(32.0 * c["SQ_INSTS_MFMA"] / elapsed, 4),
                  "valu_instructions_per_mfma": round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2),
                  "counters": {n: c[n] for n in need}}
    json.dump(out, sys.stdout, indent=1)
    print()

if __name__ == "__main__":
    main(sys.argv[1])
