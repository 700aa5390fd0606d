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
R = sys.argv[1] if len(sys.argv) > 1 else "r03"


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


def entry_for(fetch_path, write_path):
    f, w = counters(fetch_path), counters(write_path)
    byt = lambda k: int((2 * f[k][0] + w[k][0]) * 1024)
    msg = [k for k, _ in pick(f, "message_aggregate")]                     # 1 kernel at lmax 2, the degree groups above
    soft = pick(f, "attn_softmax")[0][0]
    htr = [k for k, _ in pick(f, "htr_edge")]
    gem = [k for k, _ in pick(f, "gn::gemm_") if "split" not in k]           # the projection kernels of the default mode
    n = sum(f[k][1] for k in gem)
    # message stage / message backward of one layer: all kernels of the family (general kernels, degree groups, the
    # zero-X_in kernels of the first interaction), mean over the layers
    mb = [k for k in f if "msg_bwd_" in k or "attn_bwd_kernel" in k]
    layers = f[soft][1]                       # the softmax runs once per interaction: launches = layers x steps
    entry = {
        "gn_message_aggregate": int(sum(byt(k) * f[k][1] for k in msg) / layers),
        "gn_attn_softmax": byt(soft),
        "gn_htr_edge": int(sum(byt(k) * f[k][1] for k in htr) / max(f[k][1] for k in htr)),
        "gn_gemm_family_avg": int(sum(byt(k) * f[k][1] for k in gem) / n),
        "gn_message_backward": int(sum(byt(k) * f[k][1] for k in mb) / layers) if layers else None,
        "gn_htr_backward": int(sum(byt(k) * f[k][1] for k in f if "htr_bwd_" in k) / max(1, max([f[k][1] for k in f if "htr_bwd_target" in k] or [1]))),
        "_detail": {k[:110]: {"FETCH_SIZE_KiB": f[k][0], "WRITE_SIZE_KiB": w[k][0], "launches": f[k][1]}
                    for k in msg + [soft] + htr + mb + gem},
    }
    entry["message_stage"] = entry["gn_message_aggregate"] + entry["gn_attn_softmax"]
    return entry


P = lambda name: os.path.join(ROOT, "profiles", name)
for L in (2, 4):
    res[f"lmax{L}"] = entry_for(P(f"{R}_pmc_fetch_size_lmax{L}.txt"), P(f"{R}_pmc_write_size_lmax{L}.txt"))
for wl in ("md22_ac_ala3_b64_lmax2", "md22_nanotube_b8_lmax3"):             # BASELINE configs[2], configs[4]
    if os.path.exists(P(f"{R}_pmc_fetch_size_{wl}.txt")):
        res[wl] = entry_for(P(f"{R}_pmc_fetch_size_{wl}.txt"), P(f"{R}_pmc_write_size_{wl}.txt"))
res["_round"] = R
json.dump(res, open(P("pmc_traffic.json"), "w"), indent=1)
for k, v in res.items():
    if isinstance(v, dict):
        print(k, {a: b for a, b in v.items() if a != "_detail"})
