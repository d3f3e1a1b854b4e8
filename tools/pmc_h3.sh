# SQ counters of the split-fp16 kernel family in four rocprofv3 --pmc passes (never combined with other trace domains).
#   bash tools/pmc_h3.sh          the headline kernel: 1000-proposal alanine-dipeptide flow passes on path 3  -> gpurun_out/pmc_h3_{1..4}
#   bash tools/pmc_h3.sh h1       the single-MFMA fast mode (path 4)                                       -> pmc_h1_*
#   bash tools/pmc_h3.sh 4aa      BASELINE configs[3]: wide layout, NNQQ, 512 proposals (bench.py --config 4aa) -> pmc_4aa_*
#   bash tools/pmc_h3.sh nnqq     the wide layout: NNQQ (65 atoms, 96-slot stride, three-group windows), 512 proposals  -> pmc_nnqq_*
#   bash tools/pmc_h3.sh paired   the paired 64-token layout: 100 atoms x 512 proposals (tools/time_sizes.py)           -> pmc_paired_*
#   bash tools/pmc_h3.sh dense    BASELINE configs[4]: dense softmax flow (bench.py --config dense)        -> pmc_dense_*
#   bash tools/pmc_h3.sh perop_ffn / perop_fold   the per-op split path at 256 atoms x 256 proposals: h3_ffn_tokens_kernel<4> /
#                                 attend_fold_h3_kernel (tools/time_per_op.py --path=5 256x256)            -> pmc_perop_*_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-h3}
RX=netblock_h3
case $W in
  perop_ffn) CMD="python $R/tools/time_per_op.py --path=5 256x256"; RX=h3_ffn_tokens;;
  perop_fold) CMD="python $R/tools/time_per_op.py --path=5 256x256"; RX=attend_fold_h3;;
  h3) CMD="python $R/tools/time_flow.py --iters 2 --paths 3";;
  h1) CMD="python $R/tools/time_flow.py --iters 2 --paths 4";;
  4aa) CMD="python $R/bench.py --config 4aa --steps 2 --warmup 1";;
  dense) CMD="python $R/bench.py --config dense --steps 2 --warmup 1";;
  nnqq) CMD="python $R/bench.py --config 4aa-nnqq --steps 2 --warmup 1";;
  paired) CMD="python $R/tools/time_sizes.py 100x512";;   # r05: the paired 64-token layout (97-128 atoms), reverse passes
esac
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex $RX --output-format csv -d $R/gpurun_out/pmc_${W}_$i -- $CMD > $R/gpurun_out/pmc_${W}_$i.log 2>&1
  echo "$W set $i rc=$?"
done
