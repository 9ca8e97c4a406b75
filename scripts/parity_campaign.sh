# Random-scene parity campaign on the final build (tests/test_parity_gpu.py::test_random_material_subsets...: wavefront == megakernel bit for bit,
# every fourth scene against the oracle), one line per configuration. Usage: gpurun -- 'bash scripts/parity_campaign.sh > gpurun_out/r03/parity_campaign.txt'
cd $GRAFT_REPO_ROOT
run() { # label, seeds, extra env...
  label="$1"; seeds="$2"; shift 2
  out=$(env HRPT_TEST_TRAIT_SEEDS=$seeds "$@" timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py -x -q -k random_material_subsets 2>&1 | tail -1)
  echo "$label: $seeds scenes: $out"
}
run "default" 3000
run "shadow schedule forced to the buffered query" 400 HRPT_WF_SHADOW_PATH=1
run "shadow schedule forced to ray generation + any-hit pass + resolve" 600 HRPT_WF_SHADOW_PATH=2
run "2-wide trees" 200 HRPT_WF_BVH_WIDTH=2
run "GPU PLOC builder" 200 HRPT_BVH_BUILDER=ploc
run "GPU LBVH builder" 200 HRPT_BVH_BUILDER=lbvh
run "4000-triangle scenes (global trees, overflow stacks)" 200 HRPT_TEST_TRAIT_TRIS=4000
run "4000-triangle scenes, PLOC, forced ray-generation schedule" 100 HRPT_TEST_TRAIT_TRIS=4000 HRPT_BVH_BUILDER=ploc HRPT_WF_SHADOW_PATH=2
run "4000-triangle scenes, quantised nodes forced" 200 HRPT_TEST_TRAIT_TRIS=4000 HRPT_BVH_NODE_FORMAT=2
run "4000-triangle scenes, fp32 nodes forced" 100 HRPT_TEST_TRAIT_TRIS=4000 HRPT_BVH_NODE_FORMAT=1
run "4000-triangle scenes, quantised nodes, PLOC, forced ray-generation schedule" 100 HRPT_TEST_TRAIT_TRIS=4000 HRPT_BVH_NODE_FORMAT=2 HRPT_BVH_BUILDER=ploc HRPT_WF_SHADOW_PATH=2
run "class sort forced on" 400 HRPT_WF_SHADE_SORT=1
run "class sort forced off" 400 HRPT_WF_SHADE_SORT=0
run "raygen pass kept (HRPT_WF_FUSED_PRIMARY=0)" 200 HRPT_WF_FUSED_PRIMARY=0
run "slim shadow entries off" 200 HRPT_WF_SLIM_SHADOW=0
out=$(env HRPT_TEST_TWO_LEVEL_SEEDS=300 timeout -k 10 1000 python -m pytest tests/test_two_level_gpu.py -x -q -k random_scenes 2>&1 | tail -1)
echo "two-level structure against the flat one (random instanced scenes: opaque / textured / MASK / glass / stochastic alpha, 1 or 3 lights, up to 4000 units from the origin): 300 scenes: $out"
out=$(env HRPT_TEST_TWO_LEVEL_SEEDS=150 HRPT_TLAS_BUILDER=gpu timeout -k 10 1000 python -m pytest tests/test_two_level_gpu.py -x -q -k random_scenes 2>&1 | tail -1)
echo "the same with the instance tree built on the GPU whatever the instance count (HRPT_TLAS_BUILDER=gpu): 150 scenes: $out"
out=$(env HRPT_TEST_REFIT_SEEDS=300 timeout -k 10 1000 python -m pytest tests/test_parity_gpu.py -x -q -k after_a_refit 2>&1 | tail -1)
echo "hrpt_refit_instances on random scenes (every instance moved, PLOC / LBVH hierarchies kept), wavefront and megakernel against the oracle on a fresh scene: 300 scenes: $out"
