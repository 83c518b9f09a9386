#!/usr/bin/env python3
"""profiles/rNN_limiter.json from the PMC summaries tools/profile_gpu.sh wrote (tools/prof_summary.py text): per kernel the
figures that say what limits it.  usage: limiter_json.py <round> <summary.txt> [<summary.txt> ...] > profiles/rNN_limiter.json"""
import json
import re
import sys


def counters(path):
    out = {}
    for ln in open(path):
        m = re.match(r"\s+void ptrace::(\S+?)\(.*?\s(\S+)\s+dispatches=\s*(\d+) sum=(\S+) per_dispatch=(\S+)", ln)
        if m and m.group(1).startswith("path_"):
            out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(5))
    return out


def main():
    rnd = sys.argv[1]
    res = {"round": int(rnd), "source": "rocprofv3 --pmc passes of `bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-stereo --no-reproj` "
                                        "(tools/profile_gpu.sh), per launch of the kernel; SQ_* cycle counters are in units of four cycles"}
    for path in sys.argv[2:]:
        for kernel, c in counters(path).items():
            waves_per_simd = c["SQ_WAVES"] / 1024.0
            res[kernel] = {
                "lanes_per_valu_inst": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
                "lane_fraction": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0,
                "valu_insts": c["SQ_INSTS_VALU"], "salu_insts": c["SQ_INSTS_SALU"], "lds_insts": c["SQ_INSTS_LDS"], "vmem_read_insts": c["SQ_INSTS_VMEM_RD"],
                "valu_share_of_wave_time": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                # a SIMD-32 issues a wave64 instruction in 2 of the 4 cycles the wave is charged for it
                "valu_busy": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * waves_per_simd * 0.5,
                "wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                # 256 texture addressers (one per CU); GRBM_GUI_ACTIVE is summed over the 8 XCDs
                "ta_busy": c["TA_TA_BUSY_sum"] / (256.0 * c["GRBM_GUI_ACTIVE"] / 8.0),
                "l1_accesses_per_launch": c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                "l1_to_l2_read_requests_per_launch": c["TCP_TCC_READ_REQ_sum"],
                "l2_hit_rate": c["TCC_HIT_sum"] / c["TCC_REQ_sum"],
                "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
                "waves_per_simd": waves_per_simd,
                "fetch_size_kb_per_launch": c["FETCH_SIZE"], "write_size_kb_per_launch": c["WRITE_SIZE"],
            }
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
