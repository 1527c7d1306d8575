#!/usr/bin/env python3
"""HBM bytes per launch of every kernel of the fused step, from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE, collected separately: scripts/profile_round6.sh).

Usage: scripts/make_traffic_json.py pmc_fetch_results.db pmc_write_results.db > profiles/r01_traffic.json

read  = 2 x FETCH_SIZE KiB (on gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md, HBM)
write = WRITE_SIZE KiB (uncalibrated)
The grouped-GEMM launches of a step share one kernel symbol; they are told apart by their position between two Adam
launches (plan.hip issues them in a fixed order): four per step when decoder fc1 runs as GEMMs, two (projection, weight
gradients) when the fused dec_fc1_kernel takes fc1 + squared error + dH (the default up to 5120 rows), one (projection) when
the weight gradients run on gemm_tn_kernel (T*B <= 1024)."""
import json
import sqlite3
import sys

GEMM_ORDER = {4: ["proj_gemm", "fc1_mse_gemm", "fc1_bwd_gemm", "dw_gemm"], 2: ["proj_gemm", "dw_gemm"], 1: ["proj_gemm"]}


def classify(name):
    # bf16 plans at large batches (bf16 MFMA recurrences, bf16-resident activations; keys = the plan's timer names, bench.py's `dom`)
    if "lstm_seq_bf16_kernel<false, 0" in name:
        return "enc_seq_fwd"
    if "lstm_seq_bf16_kernel<true, 0" in name:
        return "enc_seq_bwd"
    if "lstm_seq_bf16_kernel<false, 1" in name:
        return "dec_seq_fwd"
    if "lstm_seq_bf16_kernel<true, 1" in name:
        return "dec_seq_bwd"
    if "dw_stream_mixed_kernel" in name or "dw_stream_kernel" in name or "dw_reduce_kernel" in name:
        return "lstm_dw_stream"        # (two launches per step under one timer id: their bytes are summed)
    if "proj_bf16_kernel" in name:
        return "proj_gemm"
    if "dec_fc1_large" in name:
        return "fc1_mse_gemm"
    if "pack_all_kernel" in name:
        return "bf16_weight_pack"
    if "lstm_seq_small_dectail_kernel" in name:         # decoder rows + the latent chain's tail blocks (B <= 32)
        return "dec_seq_fwd"
    if "lstm_seq_small_decbwd_head_kernel" in name:     # decoder BPTT rows + the latent backward chain's head blocks (B <= 32)
        return "dec_seq_bwd"
    if "lstm_seq_small_foldproj_kernel" in name:        # ... with the projection role workgroups in front (B <= 32)
        return "enc_seq_fwd"
    if "lstm_seq_small_folddw_kernel" in name:          # ... with the weight-gradient role workgroups behind (B <= 32)
        return "enc_seq_bwd"
    if "lstm_seq_small_fold_kernel<false" in name:      # encoder recurrences + their rows' latent chains (B <= 64)
        return "enc_seq_fwd"
    if "lstm_seq_small_fold_kernel<true" in name:
        return "enc_seq_bwd"
    if "lstm_seq_small_kernel4<false" in name or "lstm_seq_fwd" in name or "lstm_seq_small_kernel<false" in name:
        return "enc_seq_fwd" if ("8, 2, 20, 30" in name or "30, 20, 8, 2" in name) else "dec_seq_fwd"
    if "lstm_seq_small_kernel4<true" in name or "lstm_seq_bwd" in name or "lstm_seq_small_kernel<true" in name:
        return "enc_seq_bwd" if ("8, 2, 20, 30" in name or "30, 20, 8, 2" in name) else "dec_seq_bwd"
    if "latent_fwd" in name:
        return "latent_fwd"
    if "latent_bwd" in name:
        return "latent_bwd"
    if "gemm_tn_kernel" in name:       # the weight gradients at T*B <= 1024 (gemm_tn.hip): then the only grouped GEMM of a step
        return "dw_gemm"               # is the projection launch
    if "gemm_f32_kernel" in name:
        return "gemm"
    if "dec_fc1_kernel" in name:
        return "fc1_mse_gemm"          # the plan's timer id of that launch (K_FC1_FWD)
    if "mse_kernel" in name:
        return "mse"
    if "adam_kernel" in name:
        return "adam"
    return None


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name=? "
                     "order by dispatch_id", (counter,)).fetchall()
    acc, step = {}, []

    def flush():
        gemms = [v for k, v in step if k == "gemm"]
        names = GEMM_ORDER.get(len(gemms))
        gi = 0
        for k, v in step:
            if k == "gemm":
                if names is None:
                    continue                      # a partial step at the start / end of the trace
                k = names[gi]
                gi += 1
            acc.setdefault(k, []).append(v)
        del step[:]

    for _, name, value in rows:
        k = classify(name)
        if k is None:
            continue
        step.append((k, value))
        if k == "adam":
            flush()
    nsteps = max(len(acc.get("adam", [])), 1)
    # (a key that several launches of a step share -- lstm_dw_stream -- is summed per step; everything else occurs once)
    return {k: sum(v) / (nsteps if k == "lstm_dw_stream" else len(v)) for k, v in acc.items()}


def main(fetch_db, write_db, workload="mosi B=32 T=20"):
    rd = per_kernel(fetch_db, "FETCH_SIZE")
    wr = per_kernel(write_db, "WRITE_SIZE")
    out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes of "
                       "bench.py, B=32 T=20; scripts/profile_round6.sh + scripts/make_traffic_json.py); read = 2 x "
                       "FETCH_SIZE KiB (gfx950 correction, MI355X_MICROARCH.md section HBM), write = WRITE_SIZE KiB "
                       "(uncalibrated)",
           "workload": workload}
    for k in sorted(set(rd) | set(wr)):
        r = int(round(2.0 * rd.get(k, 0.0) * 1024))
        w = int(round(wr.get(k, 0.0) * 1024))
        out[k] = {"read_bytes": r, "write_bytes": w, "total_bytes": r + w}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
