"""G9b: final held-out PSNR of the HIP training step over an ensemble of one-ulp-perturbed initialisations (the reference's own
ensemble is in the fixture: rerun_heldout_psnr) -- is the HIP path's distribution the reference's?"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import tests.test_hip_stage_schedule as t
import thr3ed_atom_amd as rf
from tests.helpers import *
g = load_golden("g9b_trainer_stages.npz")
G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
F = 3*(deg+1)**2; g0 = int(np.ceil(G/2)); near, far = float(g["near"]), float(g["far"]); dev = torch.device("cuda:0")
def init(seed):
    d = hash_uniform((g0,g0,g0,1),901); f = hash_uniform((g0,g0,g0,F),900+F)
    if seed:
        for arr in (d, f):
            flip = hash_uniform(arr.shape, 7000+31*seed+arr.shape[-1], 0.0, 1.0) < 0.5
            arr[...] = np.where(flip, np.nextafter(arr, np.float32(2.0)), arr)
    return t.T(d), t.T(f)
print("reference: base", float(g["heldout_psnr"]), "reruns", g["rerun_heldout_psnr"], "train0 base", float(g["train0_psnr"]), g["rerun_train0_psnr"])
for kind, storage in [("fused","split"),("fused_det","split"),("torch_optim","reference")]:
    res=[]; res0=[]
    for seed in range(int(sys.argv[1]) if len(sys.argv)>1 else 6):
        d, f = init(seed)
        grid = t.relu_grid(dev, d, f, storage)
        run = t._Follower("fused" if kind=="fused_det" else kind, dev, g); run.deterministic = kind=="fused_det"
        for stage in range(stages):
            run.start_stage(grid, g9b_learning_rate(g, stage*iters), S, near, far)
            for it in range(iters):
                step = stage*iters+it
                o,dd,px,ts,td = (t.T(a).to(dev) for a in g9b_batch(g, step))
                run.step(rf.Rays(o,dd), px, ts, td)
                if (it+1) % int(g["schedule"][2]) == 0: run.sched.step()
            run.end_stage()
            if stage == 0:
                with torch.no_grad(): grid = rf.scale_voxel_grid_with_required_output_size(grid,(G,G,G)).to(dev)
        pose = rf.CameraPose(t.T(g["heldout_rotation"]).to(dev), t.T(g["heldout_translation"]).to(dev)); intr = rf.CameraIntrinsics(hw,hw,float(g["intrinsics_stage2"][2]))
        out = run.model.render(pose,intr,perturb_sampled_points=False,num_samples_per_ray=eval_S)
        res.append(t.psnr(out.colour.cpu().numpy(), g["heldout_truth"]))
        tr = run.model.render(rf.pose_spherical(5.0,-20.0,4.0311), intr, perturb_sampled_points=False, num_samples_per_ray=eval_S)
        res0.append(t.psnr(tr.colour.cpu().numpy(), g["train0_truth"]))
    print(kind, "heldout", np.round(res,3), "mean", np.mean(res), "std", np.std(res), "| train0", np.round(res0,3), "mean", np.mean(res0))
