# round 5, call s: more end-to-end scenarios -- float16 rows through the reference's schedule (opacity reset at 3000), and the
# Gaussian-sharded scheme (the reference's own multi-GPU mode) training the demo scene on two ranks
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import importlib.util, json
spec = importlib.util.spec_from_file_location("train_demo", "tools/train_demo.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
for attr in ("f16",):
    r = m.run("engine", "default", attr, steps=3600, res=256, teacher_n=20000, student_n=20000, train_views=32, refine_start=500, refine_every=100,
              reset_every=3000, sh_interval=1000, refine_stop=15000, max_steps=30000, time_blocks=600, scene_scale=9.0)
    bm = r["loss_block_means"]
    print(attr, [b["gaussians"] for b in r["blocks"]], "before", round(min(bm[25:30]), 4), "after", round(max(bm[30:33]), 4), "end", round(bm[-1], 4),
          "psnr", round(r["psnr_heldout_before"], 2), round(r["psnr_heldout"], 2), "void", r["void_steps"])
PY
