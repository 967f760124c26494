"""is there a WELL-CONDITIONED configuration on which the bf16 build's step gradients can be held against the
f64 oracle in absolute terms? The f32 build trains `steps` steps over a small pool of batches (renderer labels: Gaussian blobs),
then ONE train step on a held-out batch of `batch` images @ `size`^2 is computed four ways on the trained variables: the f64 oracle,
the f64 oracle rounding to bf16 where the build stores bf16 (oracle.network.storage_emulation), the HIP bf16 build, the HIP f32
build. Printed per tensor: the oracle's own sensitivity to bf16 storage (emulating vs exact), the build's distance from the emulating
oracle, the cosine.   python tools/bf16_grad_bound.py [steps] [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import network as onet
from multiposenet_amd.detector.input_pipeline.heatmap_creation import get_heatmaps_batch
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer


def make_batch(seed, batch, size, persons=2):
    rs = np.random.RandomState(seed)
    h = size // 4
    people = []
    for _ in range(batch):
        kp = np.zeros((persons, 17, 3), np.int32)
        bx = np.zeros((persons, 4), np.float32)
        for p in range(persons):
            hh, ww = rs.randint(size // 3, size), rs.randint(size // 4, size // 2)
            y0, x0 = rs.randint(0, size - hh + 1), rs.randint(0, size - ww + 1)
            bx[p] = [y0, x0, y0 + hh, x0 + ww]
            kp[p, :, 0] = rs.randint(y0, y0 + hh, 17)
            kp[p, :, 1] = rs.randint(x0, x0 + ww, 17)
            kp[p, :, 2] = (rs.rand(17) < 0.8).astype(np.int32)
        people.append((kp, bx))
    heat = get_heatmaps_batch(people, size, size, 4).clone()
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + seed)
    images = torch.rand((batch, size, size, 3), generator=g, device="cuda")
    labels = {"heatmaps": heat, "loss_masks": torch.ones((batch, h, h), device="cuda"),
              "segmentation_masks": (heat.amax(-1) > 0.5).float(),
              "num_boxes": torch.full((batch,), persons, dtype=torch.int32, device="cuda")}
    return images, labels


def run(steps=300, batch=8, size=256, pool=4, lr=1e-3, verbose=True):
    hp = {"initial_learning_rate": lr, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    net = KeypointNet(dtype=torch.float32, seed=3)
    tr = Trainer(net, hp, use_graph=True)
    batches = [make_batch(s, batch, size) for s in range(pool)]
    first = last = None
    for i in range(steps):
        im, lb = batches[i % pool]
        l = tr.step({"images": im}, lb)
        if i == 0:
            first = float(l[6])
    last = float(l[6])
    trained = net.state_dict()
    del tr, net
    torch.cuda.empty_cache()
    im, lb = make_batch(99, batch, size)                  # held out: the gradient is not the ~0 of an overfit batch
    img = im.cpu().numpy()
    lab = {k: v.cpu().numpy() for k, v in lb.items()}
    ref = {k: v.astype(np.float64) for k, v in trained.items()}
    zeros = lambda: {k: np.zeros_like(v) for k, v in ref.items()}

    def oracle():
        t, _, g = onet.train_step({k: v.copy() for k, v in ref.items()}, zeros(), zeros(), img, lab, 0, hp, dtype=torch.float64)
        return t, g
    t0 = time.time()
    t_ex, g_ex = oracle()
    with onet.storage_emulation(torch.bfloat16):
        t_em, g_em = oracle()
    t_or = time.time() - t0
    keys = sorted(g_ex)
    out = {"first_loss": first, "last_loss": last, "oracle_loss": t_ex, "oracle_loss_bf16_storage": t_em, "oracle_seconds": t_or}
    res = {}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        n2 = KeypointNet(values=trained, dtype=dt)
        ls = Trainer(n2, hp, use_graph=False).step({"images": im}, lb)
        res[name] = ({k: n2.grads[k].cpu().numpy().astype(np.float64) for k in keys}, float(ls[6]))
        del n2
        torch.cuda.empty_cache()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    cos = lambda a, b: float(a.ravel() @ b.ravel() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    rows = []
    for k in keys:
        ex, em, hb, hf = (np.asarray(x[k], np.float64) for x in (g_ex, g_em, res["bf16"][0], res["f32"][0]))
        rows.append((k, ex.size, float(np.linalg.norm(ex)), rel(em, ex), rel(hb, em), cos(hb, em), rel(hb, ex), rel(hf, ex)))
    cat = lambda g: np.concatenate([np.asarray(g[k], np.float64).ravel() for k in keys])
    out.update(loss_bf16=res["bf16"][1], loss_f32=res["f32"][1], rows=rows,
               all_sens=rel(cat(g_em), cat(g_ex)), all_bf16_vs_em=rel(cat(res["bf16"][0]), cat(g_em)),
               all_bf16_vs_ex=rel(cat(res["bf16"][0]), cat(g_ex)), all_f32_vs_ex=rel(cat(res["f32"][0]), cat(g_ex)))
    if verbose:
        print(f"trained {steps} steps over {pool} batches of {batch} @ {size}^2: total loss {first:.3f} -> {last:.4f}; held-out batch: "
              f"oracle loss {t_ex:.5f}, with bf16 storage {t_em:.5f}, HIP bf16 {res['bf16'][1]:.5f}, HIP f32 {res['f32'][1]:.5f} "
              f"({t_or:.0f} s of oracle)")
        print(f"ALL gradients, relative L2: oracle's own sensitivity to bf16 storage {out['all_sens']:.4f}; HIP bf16 vs emulating oracle "
              f"{out['all_bf16_vs_em']:.4f}, vs exact oracle {out['all_bf16_vs_ex']:.4f}; HIP f32 vs exact oracle {out['all_f32_vs_ex']:.2e}")
        print("%-62s %9s %10s %8s %8s %8s %8s %9s" % ("tensor", "size", "|g|", "sens", "bf16/em", "cos", "bf16/ex", "f32/ex"))
        for r in rows:
            print("%-62s %9d %10.3e %8.4f %8.4f %8.4f %8.4f %9.2e" % r)
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    run(*a)
