#!/usr/bin/env python
"""profiles/inst_mix.json from the rocprofv3 --pmc pass "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" (+ SQ_INSTS_MFMA) of a
single-stream bench run (tools/pmc.sh <tag>_inst_mix ... with PMC_FILTER=""): per stage of a frame the vector / scalar / LDS / MFMA
instructions its kernels issue per launch.  bench.py reports the vector-ALU busy time of the dominant kernel from SQ_ACTIVE_INST_VALU (quad-cycles, x 4 / (1024 SIMDs x 2.4 GHz)) where
this pass collected it, else count x 4.2 cycles (the blend kernels' measured mix average)."""
import json
import sys

from hbm_traffic import STAGES, parse


def main(pmc_txt, out_json, gaussians, width, height):
    c = parse(pmc_txt)
    stages = {}
    for st, ks in STAGES + [("render_bwd", [("render_bwd_kernel", 1)])]:
        tot = {}
        for frag, n in ks:
            for counter in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"):
                v = [x for k, x in c.get(counter, {}).items() if frag.replace(" ", "") in k.replace(" ", "")]
                if v:
                    tot[counter] = tot.get(counter, 0.0) + n * v[0]
        stages[st] = {"valu": tot.get("SQ_INSTS_VALU"), "salu": tot.get("SQ_INSTS_SALU"), "lds": tot.get("SQ_INSTS_LDS"), "mfma": tot.get("SQ_INSTS_MFMA"),
                      "waves": tot.get("SQ_WAVES")}
        if tot.get("SQ_ACTIVE_INST_VALU"):               # quad-cycles the vector ALUs are occupied: bench.py's measured vector-ALU busy time
            stages[st]["valu_active_quad_cycles"] = tot["SQ_ACTIVE_INST_VALU"]
        if tot.get("SQ_WAVE_CYCLES"):
            stages[st]["wave_quad_cycles"] = tot["SQ_WAVE_CYCLES"]
    json.dump({"source": pmc_txt, "how": "rocprofv3 --kernel-trace --pmc passes over `python bench.py --streams 1 --exact-count`; per-launch means",
               "workload": {"gaussians": int(gaussians), "width": int(width), "height": int(height)}, "stages": stages}, open(out_json, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
