#!/bin/bash
# round 3, GPU call 1: trained-weights parity of the shipped c2f configuration in every mode (+ float32 grids in the half modes), a 10 000-iteration
# trained 8x256 NeRF (f16c margin), the torch-op trace and a kernel trace of the training iteration as the round's starting point
O=gpurun_out/r3a; mkdir -p $O
python tools/trained_c2f.py --iters 3000 --save /tmp/c2f.npz > $O/c2f_trained.log 2>&1; tail -4 $O/c2f_trained.log
EVD_F32_GRIDS=1 python tools/trained_c2f.py --load /tmp/c2f.npz --modes f16,bf16 > $O/c2f_trained_f32grids.log 2>&1; tail -2 $O/c2f_trained_f32grids.log
python tools/trained_weights.py --iters 10000 --save $O/trained_nerf_10k.npz --check > $O/nerf_10k.log 2>&1; tail -7 $O/nerf_10k.log
python tools/trace_torch_ops.py > $O/trace_ops.txt 2>&1; head -3 $O/trace_ops.txt
python tools/bench_train_step.py --iters 10 > $O/train_step.log 2>&1; tail -1 $O/train_step.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_train_step.py --iters 8 > $GRAFT_REPO_ROOT/$O/train_step_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); cp $f $O/train_step_kernel_stats.csv; rm -rf $O/prof; head -12 $O/train_step_kernel_stats.csv | cut -c1-150
