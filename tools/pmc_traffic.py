"""profiles/pmc_traffic.json from the rocprofv3 --pmc summaries of a round (tools/profile_round.sh):

    python tools/pmc_traffic.py r02

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE reports
half the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM section; checked here: WRITE_SIZE of
gn_message_aggregate = N(1+D)F*4 B and of gn_htr_edge = E*F*4 B exactly).  bench.py reads the file for `roofline.traffic`."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg=\s*([\d.]+)\s+dispatches=(\d+)", line)
        if m:
            out[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return out


def pick(tab, *needles):
    return [(k, v) for k, v in tab.items() if all(n in k for n in needles)]


res = {"_comment": __doc__.split("\n\n")[1].replace("\n", " ")}
for L in (2, 4):
    f = counters(os.path.join(ROOT, "profiles", f"{R}_pmc_fetch_size_lmax{L}.txt"))
    w = counters(os.path.join(ROOT, "profiles", f"{R}_pmc_write_size_lmax{L}.txt"))
    byt = lambda k: int((2 * f[k][0] + w[k][0]) * 1024)
    msg = [k for k, _ in pick(f, "message_aggregate")]                     # 1 kernel at lmax 2, the degree groups above
    soft = pick(f, "attn_softmax_kernel")[0][0]
    htr = pick(f, "htr_edge_kernel")[0][0]
    gem = [k for k, _ in pick(f, "gn::gemm_") if "split" not in k]           # the projection kernels of the default mode
    n = sum(f[k][1] for k in gem)
    entry = {
        "gn_message_aggregate": sum(byt(k) for k in msg),
        "gn_attn_softmax": byt(soft),
        "gn_htr_edge": byt(htr),
        "gn_gemm_family_avg": int(sum(byt(k) * f[k][1] for k in gem) / n),
        "_detail": {k[:110]: {"FETCH_SIZE_KiB": f[k][0], "WRITE_SIZE_KiB": w[k][0], "launches": f[k][1]}
                    for k in msg + [soft, htr] + gem},
    }
    entry["message_stage"] = entry["gn_message_aggregate"] + entry["gn_attn_softmax"]
    res[f"lmax{L}"] = entry
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
for L in (2, 4):
    print(L, {k: v for k, v in res[f"lmax{L}"].items() if k != "_detail"})
