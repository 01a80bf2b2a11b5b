# round 5, call t: does a 360-degree (equirectangular) scene train?  the reference's default camera model, cameras inside the cloud
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import importlib.util, json
spec = importlib.util.spec_from_file_location("train_demo", "tools/train_demo.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
for path, strategy in (("engine", "default"), ("operator", "default"), ("engine", "mcmc")):
    r = m.run(path, strategy, steps=800, res=128, train_views=32, camera_model="spherical")
    bm = r["loss_block_means"]
    print(path, strategy, "N", r["n_final"], "psnr", round(r["psnr_heldout_before"], 2), "->", round(r["psnr_heldout"], 2), "ssim", round(r["ssim_heldout"], 4),
          "loss", [round(x, 4) for x in bm], "void", r["void_steps"], "engine", r["fused_engine_ran"])
PY
