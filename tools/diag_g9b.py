import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import tests.test_hip_stage_schedule as t
import thr3ed_atom_amd as rf
from tests.helpers import *
g = load_golden("g9b_trainer_stages.npz")
G, deg, hw, n_img, n_rays, iters, S, stages, eval_S, seed0 = (int(v) for v in g["config"])
F = 3*(deg+1)**2; g0 = int(np.ceil(G/2)); near, far = float(g["near"]), float(g["far"]); dev = torch.device("cuda:0")
for kind, storage in [("fused","split"),("autograd","reference"),("torch_optim","reference")]:
    grid = t.relu_grid(dev, t.T(hash_uniform((g0,g0,g0,1),901)), t.T(hash_uniform((g0,g0,g0,F),900+F)), storage)
    run = t._Follower(kind, dev, g); rel=[]
    for stage in range(stages):
        run.start_stage(grid, g9b_learning_rate(g, stage*iters), S, near, far)
        for it in range(iters):
            step = stage*iters+it
            o,d,px,ts,td = (t.T(a).to(dev) for a in g9b_batch(g, step))
            ls, ld = run.step(rf.Rays(o,d), px, ts, td)
            rel.append(max(abs(ls/g["specular_loss"][step]-1), abs(ld/g["diffuse_loss"][step]-1)))
            if (it+1) % int(g["schedule"][2]) == 0: run.sched.step()
            if step+1 in list(g["checkpoints"]):
                k = list(g["checkpoints"]).index(step+1)
                pose = rf.CameraPose(t.T(g["heldout_rotation"]).to(dev), t.T(g["heldout_translation"]).to(dev)); intr = rf.CameraIntrinsics(hw,hw,float(g["intrinsics_stage2"][2]))
                out = run.model.render(pose,intr,perturb_sampled_points=False,num_samples_per_ray=eval_S)
                print(kind, "checkpoint", step+1, "psnr ours", t.psnr(out.colour.cpu().numpy(), g["heldout_truth"]), "ref", float(g["checkpoint_heldout_psnr"][k]), "reruns-base", (g["rerun_checkpoint_heldout_psnr"][:,k]-g["checkpoint_heldout_psnr"][k]).round(4), "img maxabs", np.abs(out.colour.cpu().numpy()-g["checkpoint_heldout_render"][k]).max())
        run.end_stage()
        if stage == 0:
            dd = np.abs(grid.densities.detach().cpu().numpy()-g["dens_stage1_end"]); df = np.abs(grid.features.detach().cpu().numpy()-g["feat_stage1_end"])
            print(kind, "stage1 end: dens within 1e-4/1e-3/5e-3/5e-2:", [float(np.mean(dd<x)) for x in (1e-4,1e-3,5e-3,5e-2)], "feat:", [float(np.mean(df<x)) for x in (1e-4,1e-3,5e-3,5e-2)], "max", dd.max(), df.max())
            with torch.no_grad(): grid = rf.scale_voxel_grid_with_required_output_size(grid,(G,G,G)).to(dev)
    dd = np.abs(grid.densities.detach().cpu().numpy()-g["dens_final"]); df = np.abs(grid.features.detach().cpu().numpy()-g["feat_final"])
    print(kind, "final: dens", [float(np.mean(dd<x)) for x in (1e-4,1e-3,5e-3,5e-2)], "feat", [float(np.mean(df<x)) for x in (1e-4,1e-3,5e-3,5e-2)])
    r = np.array(rel); print(kind, "loss rel median per 50:", [float(np.median(r[i:i+50])) for i in range(0,len(r),50)]); print(kind, "loss rel: first10", r[:10].max(), "median", np.median(r), "p90", np.percentile(r,90), "max", r.max(), "argmax", r.argmax(), "stage2 first3", r[iters:iters+3])
    pose = rf.CameraPose(t.T(g["heldout_rotation"]).to(dev), t.T(g["heldout_translation"]).to(dev)); intr = rf.CameraIntrinsics(hw,hw,float(g["intrinsics_stage2"][2]))
    out = run.model.render(pose,intr,perturb_sampled_points=False,num_samples_per_ray=eval_S)
    print(kind, "heldout psnr ours", t.psnr(out.colour.cpu().numpy(), g["heldout_truth"]), "ref", float(g["heldout_psnr"]), "max abs img diff", np.abs(out.colour.cpu().numpy()-g["heldout_render"]).max())
    tr = run.model.render(rf.pose_spherical(5.0,-20.0,4.0311), intr, perturb_sampled_points=False, num_samples_per_ray=eval_S)
    print(kind, "train0 psnr ours", t.psnr(tr.colour.cpu().numpy(), g["train0_truth"]), "ref", float(g["train0_psnr"]))
