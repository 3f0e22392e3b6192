#!/usr/bin/env python
"""profiles/inst_mix.json from the rocprofv3 --pmc pass "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" (+ SQ_INSTS_MFMA) of a
single-stream bench run (tools/pmc.sh <tag>_inst_mix ... with PMC_FILTER=""): per stage of a frame the vector / scalar / LDS / MFMA
instructions its kernels issue per launch.  bench.py turns the vector count into the time the chip's 1024 SIMDs need to ISSUE them
(a wave64 vector instruction occupies its SIMD for 4 cycles) - the ceiling the blend kernels run against."""
import json
import sys

from hbm_traffic import STAGES, parse


def main(pmc_txt, out_json, gaussians, width, height):
    c = parse(pmc_txt)
    stages = {}
    for st, ks in STAGES:
        tot = {}
        for frag, n in ks:
            for counter in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_WAVES"):
                v = [x for k, x in c.get(counter, {}).items() if frag.replace(" ", "") in k.replace(" ", "")]
                if v:
                    tot[counter] = tot.get(counter, 0.0) + n * v[0]
        stages[st] = {"valu": tot.get("SQ_INSTS_VALU"), "salu": tot.get("SQ_INSTS_SALU"), "lds": tot.get("SQ_INSTS_LDS"), "mfma": tot.get("SQ_INSTS_MFMA"),
                      "waves": tot.get("SQ_WAVES")}
    json.dump({"source": pmc_txt, "how": "rocprofv3 --kernel-trace --pmc passes over `python bench.py --streams 1 --exact-count`; per-launch means",
               "workload": {"gaussians": int(gaussians), "width": int(width), "height": int(height)}, "stages": stages}, open(out_json, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
