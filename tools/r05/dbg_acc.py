import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import cases
from oracle import ref_np as R
from samplenerfro_amd import models, ops, prng, synthetic as syn, utils as U
from samplenerfro_amd.train import TrainState, train_step
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
img = np.load(os.path.join(ROOT, "tests", "golden", "example_image.npz"))["rgba_sum4"]
pixels = (img[..., :3].astype(np.float32) / np.float32(1020.0)).reshape(-1, 3)
_, _, counts = cases.load_example_obj()
grid = cases.example_grid(counts).astype(np.float32)
H = W = 400
focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
o, _, v = R.generate_rays(cases.EXAMPLE_C2W, H, W, focal=focal)
flags = U.default_flags(num_coarse_samples=64, num_fine_samples=128, num_path_samples=12, white_bkgd=False, use_online_sparsity=False, randomized=True, near=2.0, far=6.0,
                        batch_size=1024, bg_weight=0.025, bg_smooth_weight=1.0, bg_patch_size=128, lr_init=5e-4, lr_final=5e-6, lr_delay_steps=0, max_steps=30000, config="configs/example")
model, variables = models.construct_nerf(np.array([0, 7], np.uint32), None, flags, [128] * 3, [-1.5] * 3, [1.5] * 3, T(grid))
pf = syn.init_params_flat(7, fine=True)
for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp"):
    variables["flat"][k].copy_(T(pf[k]))
state = TrainState.create(model, variables, flags)
o_d, v_d, pix_d = T(o.reshape(-1, 3)), T(v.reshape(-1, 3)), T(pixels)
rays_hw = U.Rays(o_d.reshape(H, W, 3), None, v_d.reshape(H, W, 3), None)
fn = lambda k0, k1, rays, path=None: model.apply(state.variables, k0, k1, rays, False, path=path)
gen = np.random.default_rng(1); ev = R.safe_l2_normalize(gen.standard_normal((128, 128, 3)).astype(np.float32)); env = U.Rays(None, None, T(ev), None)
pick = torch.Generator(device="cpu").manual_seed(5); r = prng.PRNGKey(7)
for i in range(0, 301):
    if i in (0, 20, 60, 150, 300):
        rgb, dist, acc = U.render_image(fn, rays_hw, prng.PRNGKey(1), False, chunk=8192)
        print(f"step {i}: acc mean {float(acc.mean()):.4e} max {float(acc.max()):.4e}  psnr {U.compute_psnr(float(((rgb.reshape(-1,3)-pix_d)**2).mean())):.2f}", flush=True)
    idx = torch.randint(0, H * W, (1024,), generator=pick).to("cuda:0")
    batch = {"rays": U.Rays(o_d[idx], None, v_d[idx], None), "pixels": pix_d[idx], "annealed_alpha": (i + 1) / 160000.0, "env_rays": env}
    state, stats, r = train_step(model, r, state, batch, flags)
