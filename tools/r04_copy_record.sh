#!/bin/bash
# copies the artefacts of tools/r04_final.sh from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
R=${1:-5696}
cp gpurun_out/r04_final_bench_default.json profiles/r04_final_bench_default.json
cp gpurun_out/r04_final_c1_trace.txt profiles/r04_final_c1_kernel_trace.txt
cp gpurun_out/r04_final_c1_pmc_mfma_lds.txt profiles/r04_final_c1_pmc_mfma_lds.txt
cp gpurun_out/r04_final_c1_gaps.txt profiles/r04_final_c1_step_gaps.txt
cp gpurun_out/r04_final_c1_seq.txt profiles/r04_final_c1_step_sequence.txt
cp gpurun_out/r04_final_c3_trace.txt profiles/r04_final_c3_kernel_trace.txt
cp gpurun_out/r04_final_c4_trace.txt profiles/r04_final_c4_kernel_trace.txt
cp gpurun_out/r04_gu_pmc.json profiles/r04_gu_geglu_gemm_pmc_T$R.json
cp gpurun_out/r04_wgrad_pmc.json profiles/r04_wgrad_gemm_pmc_T$R.json
cp gpurun_out/r04_other_workloads.json profiles/r04_other_workloads.json
