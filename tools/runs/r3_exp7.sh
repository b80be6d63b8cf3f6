#!/bin/bash
# round 3, GPU call 7: AWP per-ray tail as a captured graph; PMC passes of the headline MLP kernel, the gathers (f32 / f16 grids) and the new scatter
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_gpu_awp.py -x -q -m gpu > $O/test_awp.log 2>&1; tail -5 $O/test_awp.log
python tools/bench_train_step.py --iters 10 --awp fused 2>&1 | tail -1 | tee $O/train_step_awp.log
python tools/bench_train_step.py --iters 10 --awp fused --no-graph 2>&1 | tail -1 | tee -a $O/train_step_awp.log
bash tools/pmc_mlp.sh f16c,f16 $O/pmc_mlp > $O/pmc_mlp_summary.txt 2>&1; python tools/pmc_mlp_json.py $O/pmc_mlp $O/pmc_mlp.json | tail -12
cd $GRAFT_REPO_ROOT
bash tools/pmc_voxel.sh $O/pmc_voxel_f16c f16c > $O/pmc_voxel_f16c.txt 2>&1; tail -30 $O/pmc_voxel_f16c.txt
cd $GRAFT_REPO_ROOT
bash tools/pmc_scatter.sh $O/pmc_scatter 0.05 > $O/pmc_scatter.txt 2>&1; tail -40 $O/pmc_scatter.txt
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" -size +2M -delete; du -sh $O
