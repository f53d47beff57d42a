#!/usr/bin/env python3
"""Summarises rocprofv3 PMC passes of `bench.py` into profiles/<tag>_pmc_summary.json and
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES \
              SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d DIR_sq -- python bench.py ...
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d DIR_write -- python bench.py ...
    python tools/pmc_summary.py TAG DIR_sq DIR_fetch DIR_write

Corrections (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE
is exact for 16-byte-per-lane stores.  Every kernel here reads and writes 16 B per lane.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name: str) -> str:
    """rocprofv3 symbol -> the kernel family name bench.py / the engine use (one symbol per family)."""
    m = re.match(r"(?:void )?(?:hmv::)?conv_igemm<(float|_Float16), (\d+), (\d+), \d+, \d+, (\d), (?:false|true), (\d+)(?:, (false|true))?(?:, (false|true))?(?:, (?:false|true))?(?:, (false|true))?>", name)
    if m:
        t = "f32" if m.group(1) == "float" else "f16"
        k16 = ",k16" if (t == "f32" and m.group(5) == "16") else ""
        skip = ",rowsum" if m.group(7) == "true" else (",skipN" if m.group(6) == "true" else "")
        skip += ",c32" if m.group(8) == "true" else ""
        return f"conv_igemm_{t}<{m.group(2)}x{m.group(3)}{k16}," + {"0": "taps", "1": "1x1", "2": "dense", "3": "halo"}[m.group(4)] + skip + ">"
    # rocprofv3 leaves the _Float16 instantiations mangled (DF16_): conv_igemm<_Float16, BM, BN, WGM, WGN, MODE, GENERIC, KB, PARTN, RD>
    m = re.match(r"_ZN3hmv10conv_igemmIDF16_Li(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d)ELb[01]ELi(\d+)ELb[01]ELb([01])E(?:Lb[01]ELb([01])E)?", name)
    if m:
        c32 = m.group(6) == "1"   # the 32-channel-chunk K order of the layers packed for conv_ht.hip
        k16 = ",k16" if (m.group(4) == "32" and not c32) else ""
        return f"conv_igemm_f16<{m.group(1)}x{m.group(2)}{k16}," + {"0": "taps", "1": "1x1", "2": "dense", "3": "halo"}[m.group(3)] + \
            (",rowsum" if m.group(5) == "1" else "") + (",c32" if c32 else "") + ">"
    # round 3: conv_stream_f16<TM, TN, MW, NW, NP, NSLOT, HAS_RES, SPREAD, NT, DUAL, N2> and conv_gemm8_f16<DUAL>
    m = re.match(r"(?:void )?(?:hmv::)?conv_stream_f16<(\d+), (\d+), (\d+), (\d+), (\d+), \d+, (true|false)", name)
    if m:
        tm, tn, mw, nw, np_ = (int(m.group(i)) for i in range(1, 6))
        targs = name[name.index("<") + 1:name.index(">")].split(", ")
        tail = ",res" if m.group(6) == "true" else (",dual" if len(targs) >= 10 and targs[9] == "true" else "")
        if len(targs) >= 11 and targs[10] != "0":   # N2: the chained 1x1 conv (conv3 -> the next Bottleneck's conv1)
            tail += f",+1x1:{targs[10]}"
        return f"conv_stream_f16<{32 * tm * mw}x{32 * tn * nw},k{64 * np_}" + tail + ">"
    # conv_stream_f32<TM, TN, MW, NW, NP, NSLOT, HAS_RES, DUAL, HALF>: 32-channel pieces
    m = re.match(r"(?:void )?(?:hmv::)?conv_stream_f32<(\d+), (\d+), (\d+), (\d+), (\d+), \d+, (true|false)(?:, (true|false))?(?:, (?:true|false))?>", name)
    if m:
        tm, tn, mw, nw, np_ = (int(m.group(i)) for i in range(1, 6))
        tail = ",res>" if m.group(6) == "true" else (",dual>" if m.group(7) == "true" else ">")
        return f"conv_stream_f32<{32 * tm * mw}x{32 * tn * nw},k{32 * np_}" + tail
    # conv_hs_f16<R, S, CPP, TM, TN, MW, NW, NSLOT, HAS_RES, POOL>: CPP 16-byte chunks (8 channels each) per pixel; 40-channel layers keep 40 outputs
    m = re.match(r"(?:void )?(?:hmv::)?conv_hs_f16<(\d+), (\d+), (\d+), \d+, (\d+), \d+, (\d+), \d+, (true|false)(?:, (true|false))?>", name)
    if m:
        r_, s_, cpp, tn, nw = (int(m.group(i)) for i in range(1, 6))
        cout = 40 if cpp == 5 else (80 if cpp == 10 else 32 * tn * nw)   # 40- / 80-channel layers run on 64 / 96 weight rows
        return f"conv_hs_f16<{r_}x{s_},{8 * cpp}->{cout}" + (",res" if m.group(6) == "true" else "") + (",+maxpool>" if m.group(7) == "true" else ">")
    m = re.match(r"(?:void )?(?:hmv::)?conv_rds_f32<(\d+), (true|false), \d+>", name)
    if m:
        return f"conv_rds_f32<3x3,{m.group(1)}->{m.group(1)}" + (",res>" if m.group(2) == "true" else ">")
    if re.match(r"(?:void )?(?:hmv::)?conv_hs_stem_f32<", name):
        return "conv_hs_stem_f32<4x4,12->64>"
    # round 4: conv_ht_f16<M16>, conv_gemm8_f16<DUAL, M16>, conv_m16_f16<BM, BN, TAPS, DUAL>, gemm_x3_f16<BM, BN, WGM, WGN> (demangled or not)
    m = re.match(r"(?:void )?(?:hmv::)?conv_gemm8p_f16<(true|false)>", name) or re.match(r"_ZN3hmv15conv_gemm8p_f16ILb([01])EE", name)
    if m:
        return "conv_gemm8_f16<256x256,1x1," + ("dual," if m.group(1) in ("true", "1") else "") + "m16,persistent>"
    if re.match(r"(?:void )?(?:hmv::)?conv_htp_f16\(", name) or name.startswith("_ZN3hmv12conv_htp_f16E"):
        return "conv_ht_f16<512x128,3x3,m16,persistent>"
    m = re.match(r"(?:void )?(?:hmv::)?conv_ht_f16<(true|false)>", name) or re.match(r"_ZN3hmv11conv_ht_f16ILb([01])EE", name)
    if m:
        return "conv_ht_f16<512x128,3x3" + (",m16>" if m.group(1) in ("true", "1") else ">")
    m = re.match(r"(?:void )?(?:hmv::)?conv_gemm8_f16<(true|false)(?:, (true|false))?>", name) or re.match(r"_ZN3hmv14conv_gemm8_f16ILb([01])ELb([01])EE", name)
    if m:
        dual, m16 = m.group(1) in ("true", "1"), m.group(2) in ("true", "1")
        return "conv_gemm8_f16<256x256,1x1" + (",dual" if dual else "") + (",m16>" if m16 else ">")
    m = re.match(r"(?:void )?(?:hmv::)?conv_m16_f16<(\d+), (\d+), (true|false)(?:, (true|false))?>", name) or \
        re.match(r"_ZN3hmv12conv_m16_f16ILi(\d+)ELi(\d+)ELb([01])ELb([01])EE", name)
    if m:
        taps, dual = m.group(3) in ("true", "1"), m.group(4) in ("true", "1")
        return f"conv_m16_f16<{m.group(1)}x{m.group(2)}," + ("taps,c32>" if taps else ("1x1,dual>" if dual else "1x1>"))
    m = re.match(r"(?:void )?(?:hmv::)?gemm_x3_f16<(\d+), (\d+), \d+, \d+>", name) or re.match(r"_ZN3hmv11gemm_x3_f16ILi(\d+)ELi(\d+)E", name)
    if m:
        return f"gemm_x3_f16<{m.group(1)}x{m.group(2)}>"
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("hmv::", "")


def load(d):
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(max(files, key=os.path.getmtime))))   # the newest pass if older ones were merged into the directory


def main():
    tag, d_sq, d_fetch, d_write = sys.argv[1:5]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in load(d_sq):
        f = family(r["Kernel_Name"])
        agg[f][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[f] += 1
            agg[f]["dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    sq = {}
    for f, c in agg.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0          # the counter sums the 8 XCDs
        if not cyc or not c["SQ_WAVE_CYCLES"]:
            continue
        sq[f] = {"launches": n[f], "total_ms": round(c["dur_ns"] / 1e6, 3),
                 "effective_clock_ghz": round(cyc / (c["dur_ns"] * 1e-9) / 1e9, 3),
                 "mfma_busy_frac": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),   # 1024 SIMDs
                 "wave_parked_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                 "wave_issue_stall_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                 "wave_active_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4),
                 "lds_bank_conflict_per_wave_cycle": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_WAVE_CYCLES"], 5)}
    traffic = {}
    for d, cn, corr in ((d_fetch, "FETCH_SIZE", 2.0), (d_write, "WRITE_SIZE", 1.0)):
        a, k = collections.defaultdict(float), collections.Counter()
        for r in load(d):
            if r["Counter_Name"] == cn:
                f = family(r["Kernel_Name"])
                a[f] += float(r["Counter_Value"]) * 1024.0 * corr
                k[f] += 1
        for f in a:
            traffic.setdefault(f, {})[cn.lower() + "_bytes_per_launch"] = a[f] / k[f]
            traffic[f]["launches"] = k[f]
    for f, t in traffic.items():
        t["hbm_bytes_per_launch"] = t.get("fetch_size_bytes_per_launch", 0.0) + t.get("write_size_bytes_per_launch", 0.0)
    out = {"tag": tag, "sq": dict(sorted(sq.items(), key=lambda kv: -kv[1]["total_ms"])), "traffic": traffic,
           "corrections": "FETCH_SIZE x2 (gfx950 wide loads), KiB -> bytes; WRITE_SIZE exact"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
    if "--no-traffic-file" not in sys.argv:   # profiles/pmc_traffic.json: one entry per arithmetic mode of bench.py (--dtype)
        dtype = "f32"
        for a in sys.argv:
            if a.startswith("--dtype="):
                dtype = a.split("=", 1)[1]
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            allm = json.load(open(path))
            if "kernels" in allm:      # round-1 layout: one unnamed (fp32) entry
                allm = {"f32": dict(allm, dtype="f32")}
        except (OSError, ValueError):
            allm = {}
        allm[dtype] = {"source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --dtype {dtype})",
                       "dtype": dtype, "kernels": {f: t["hbm_bytes_per_launch"] for f, t in traffic.items()}}
        json.dump(allm, open(path, "w"), indent=1)
    for f, v in list(out["sq"].items())[:6]:
        print(f, v, {k: round(x / 1e6, 1) for k, x in traffic.get(f, {}).items() if k.endswith("launch")})


if __name__ == "__main__":
    main()
