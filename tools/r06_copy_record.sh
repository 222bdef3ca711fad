#!/bin/bash
# copies the artefacts of tools/r06_final.sh from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
R=${1:-5696}
cp gpurun_out/r06_final_bench_default.json profiles/r06_final_bench_default.json
cp gpurun_out/r06_final_c1_trace.txt profiles/r06_final_c1_kernel_trace.txt
cp gpurun_out/r06_final_c1_pmc_mfma_lds.txt profiles/r06_final_c1_pmc_mfma_lds.txt
cp gpurun_out/r06_final_c1_gaps.txt profiles/r06_final_c1_step_gaps.txt
cp gpurun_out/r06_final_c1_seq.txt profiles/r06_final_c1_step_sequence.txt
cp gpurun_out/r06_final_c3_trace.txt profiles/r06_final_c3_kernel_trace.txt
cp gpurun_out/r06_final_c4_trace.txt profiles/r06_final_c4_kernel_trace.txt
cp gpurun_out/r06_gu_pmc.json profiles/r06_gu_geglu_gemm_pmc_T$R.json
cp gpurun_out/r06_wgrad_pmc.json profiles/r06_wgrad_gemm_pmc_T$R.json
cp gpurun_out/r06_other_workloads.json profiles/r06_other_workloads.json
cp gpurun_out/r06_attn_oproj_pmc.json profiles/r06_attn_oproj_pmc.json
cp gpurun_out/r06_attn_oproj_bench.txt profiles/r06_attn_oproj_bench.txt
cp gpurun_out/r06_dh_pmc.json profiles/r06_dh_geglu_bwd_gemm_pmc_T$R.json 2>/dev/null
cp gpurun_out/parity_errors.json profiles/r06_parity_errors.json 2>/dev/null
cp gpurun_out/r06_dxn2_pmc.json profiles/r06_dxn2_gemm_pmc_T$R.json 2>/dev/null
cp gpurun_out/r06_attn_oproj_bench_S40.txt profiles/r06_attn_oproj_bench_S40.txt 2>/dev/null
cp gpurun_out/r06_attn_oproj_bench_S56.txt profiles/r06_attn_oproj_bench_S56.txt 2>/dev/null
cp gpurun_out/r06_final_c1_S40_trace.txt profiles/r06_final_c1_S40_kernel_trace.txt 2>/dev/null
cp gpurun_out/r06_final_c1_S40_seq.txt profiles/r06_final_c1_S40_step_sequence.txt 2>/dev/null
