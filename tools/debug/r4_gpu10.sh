cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4i
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4i/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4i/bench_k20.json 2> gpurun_out/r4i/bench_k20.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4i/bench_k20.json"))
print(d["value"], d["value_all_pairs"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
c = d["secondary"]["configs"]
print("c2", c["config2_latent_knn_b256"]["attack_ms_per_iteration"], c["config2_latent_knn_b256"]["knn_dists_k8_ms"], c["config2_latent_knn_b256"]["defend_surface_ms"], c["config2_latent_knn_b256"]["roofline_knn"]["frac"])
print("c3", c["config3_chamfer_emd_b128"]["attack_ms_per_iteration"], c["config3_chamfer_emd_b128"]["roofline_emd"])
print("c4", c["config4_n8192_b32"]["grid_search"], c["config4_n8192_b32"]["roofline_chamfer"]["frac"], c["config4_n8192_b32"]["roofline_chamfer"]["traffic_over_algorithmic"])
print("tv", d["secondary"]["trained_victim"]["grid_search"]["attack_iterations_per_sec"], d["secondary"]["trained_victim"]["all_pairs"], d["secondary"]["trained_victim"]["mean_fraction_handed_back"])
print("emd", d["secondary"]["roofline_emd"]["B32"], d["secondary"]["roofline_emd"]["B128"])
print("train", d["secondary"]["ae_training_step"])
PY
