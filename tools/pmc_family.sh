#!/bin/bash
# Round 4, VERDICT r3 item 1: SQ counters + HBM traffic of the strided / 1x1 convolution family at HRNet shapes (conv_microbench indices).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04pmc
mkdir -p $O
export FS_CONV_PRECISION=bf16x3
cd $R
for spec in "fwd 7 conv_tapset_kernel" "bwd_data 7 conv_tapset_kernel" "fwd 11 conv_tapset_kernel" "fwd 5 conv1x1_gemm_kernel" "fwd 6 conv1x1_gemm_kernel" \
            "bwd_data 5 conv1x1_gemm_kernel" "wgrad 7 conv_wgrad_class_kernel" "wgrad 5 linear_wgrad_kernel" "fwd 9 conv_igemm_split_kernel" "wgrad 11 conv_wgrad_class_kernel"; do
  set -- $spec
  bash tools/pmc_conv.sh $1 $2 $3 > $O/sq_bf16x3_$3_$1_shape$2.txt 2>&1
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_t
    rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_t --output-format csv -- python3 $R/tools/conv_microbench.py $1 3 $2 > $R/gpurun_out/pmc_t.log 2>&1
    python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t $3 >> $O/sq_bf16x3_$3_$1_shape$2.txt
  done
  cd $R
  echo "[pmc] $spec done"
done
rm -rf $R/gpurun_out/pmc_1 $R/gpurun_out/pmc_2 $R/gpurun_out/pmc_3 $R/gpurun_out/pmc_t
