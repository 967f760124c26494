"""The bf16 build's BACKWARD pass at the ORACLE's forward state ("teacher forcing"). tools/bf16_grad_bound.py
shows why a plain comparison of step gradients cannot carry an absolute bound: on trained variables and a held-out batch of 8 @ 256^2
the f64 oracle's own gradient moves by 0.93 in relative L2 when it rounds where the build stores bf16 (3 of 127 tensors below 0.05) -
the forward perturbation flips activation masks and shifts batch statistics, and the backward pass amplifies that. Here the
perturbation is taken out: the bf16 build runs its forward pass (sizing every buffer), then every tensor its backward pass reads -
raw conv outputs, FPN sums, concat slices, logits, and the batch statistics / affines of all 40 batch-norm layers - is OVERWRITTEN with
the emulating oracle's (exactly representable) values, and the build's loss gradient + backward pass run from there. What remains
is the arithmetic of the backward kernels themselves: bf16 gradient storage, f32 accumulation order, the fused reductions.
python tools/bf16_teacher_forced.py [train steps] [batch] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import network as onet
from multiposenet_amd.net import KeypointNet
from multiposenet_amd.train import Trainer

BN_EPS = 1e-3


def labels_np(rs, B, h, w):
    hm = (rs.rand(B, h, w, 17) * 0.2).astype(np.float32)
    for b in range(B):
        for _ in range(12):
            y, x, c = rs.randint(1, h - 1), rs.randint(1, w - 1), rs.randint(17)
            hm[b, y - 1:y + 2, x - 1:x + 2, c] = 0.6
            hm[b, y, x, c] = 1.0
    return {"heatmaps": hm, "loss_masks": (rs.rand(B, h, w) < 0.9).astype(np.float32),
            "segmentation_masks": (rs.rand(B, h, w) < 0.3).astype(np.float32), "num_boxes": rs.randint(1, 4, B).astype(np.int32)}


def trained_variables(steps, B, size, seed=5, pool=3, lr=1e-3):
    """the f32 build, `steps` steps over a small pool of batches: variables, batch-norm parameters and moving statistics of a network
    that has left its initialisation (nothing here depends on HOW well it is trained)"""
    hp = {"initial_learning_rate": lr, "num_steps": 200000, "weight_decay": 0.0, "depth_multiplier": 1.0}
    net = KeypointNet(dtype=torch.float32, seed=seed)
    tr = Trainer(net, hp, use_graph=False)
    rs = np.random.RandomState(seed)
    batches = []
    for _ in range(pool):
        img = torch.tensor(rs.rand(B, size, size, 3).astype(np.float32)).cuda()
        lab = {k: torch.tensor(v).cuda() for k, v in labels_np(rs, B, size // 4, size // 4).items()}
        batches.append((img, lab))
    first = last = None
    for i in range(steps):
        l = tr.step({"images": batches[i % pool][0]}, batches[i % pool][1])
        first = float(l[6]) if i == 0 else first
    last = float(l[6]) if steps else None
    return net.state_dict(), first, last


def run(steps=60, B=4, size=128, seed=9, dtype=torch.bfloat16, verbose=True):
    values, first, last = trained_variables(steps, B, size)
    rs = np.random.RandomState(seed)
    img = rs.rand(B, size, size, 3).astype(np.float32)                 # a batch the variables have not seen
    lab = labels_np(rs, B, size // 4, size // 4)
    # ---- the emulating oracle: forward with every stored tensor tapped, backward
    em = torch.bfloat16 if dtype == torch.bfloat16 else None
    p64 = {k: torch.tensor(v, dtype=torch.float64, requires_grad=onet.is_trainable(k)) for k, v in values.items()}
    taps = {}
    ctx = onet.storage_emulation(em) if em is not None else None
    if ctx:
        ctx.__enter__()
    try:
        heat, enr = onet.forward(torch.tensor(img, dtype=torch.float64), p64, True, taps=taps)
        tl = {k: torch.tensor(v) if k == "num_boxes" else torch.tensor(v, dtype=torch.float64) for k, v in lab.items()}
        total, _ = onet.losses_fn(heat, enr, tl)
        total.backward()
    finally:
        if ctx:
            ctx.__exit__(None, None, None)
    want = {k: p64[k].grad.numpy() for k in values if onet.is_trainable(k)}
    # ---- the build: forward (buffers), then the oracle's forward state in its place
    net = KeypointNet(values=values, dtype=dtype)
    net.forward(torch.tensor(img).cuda(), True)
    b = net._last[0]

    def nhwc(t):
        return t.detach().permute(0, 2, 3, 1).contiguous()

    def put(dst, src_nhwc):
        assert tuple(dst.shape) == tuple(src_nhwc.shape), (dst.shape, src_nhwc.shape)
        dst.copy_(src_nhwc.to(dst.dtype).cuda())

    def put_bn(bn, raw_nchw):
        x = raw_nchw.detach()
        mean, var = x.mean(dim=(0, 2, 3)), x.var(dim=(0, 2, 3), unbiased=False)
        invstd = torch.rsqrt(var + BN_EPS)
        gamma, beta = p64[bn.name + "/gamma"].detach(), p64[bn.name + "/beta"].detach()
        scale = gamma * invstd
        for dst, src in ((bn.mean, mean), (bn.invstd, invstd), (bn.scale, scale), (bn.shift, beta - mean * scale)):
            dst.copy_(src.float().cuda())

    put(b["stem"], nhwc(taps["MobilenetV1/Conv2d_0/raw"]))
    put_bn(net.stem_bn, taps["MobilenetV1/Conv2d_0/raw"])
    for i, blk in enumerate(net.blocks):
        d, p = f"MobilenetV1/Conv2d_{i + 1}_depthwise/raw", f"MobilenetV1/Conv2d_{i + 1}_pointwise/raw"
        put(b["dw"][i], nhwc(taps[d])); put_bn(blk["dw_bn"], taps[d])
        put(b["pw"][i], nhwc(taps[p])); put_bn(blk["pw_bn"], taps[p])
    for l in (2, 3, 4, 5):
        put(b["x"][l], nhwc(taps[f"x{l}"]))
        put(b["p"][l], enr[f"p{l}"].detach()); put_bn(net.p_bn[l], enr[f"p{l}"].detach().permute(0, 3, 1, 2))
        put(b["y1"][l], nhwc(taps[f"phi_subnet_{l}/y1"])); put_bn(net.phi[l]["bn1"], taps[f"phi_subnet_{l}/y1"])
        put(b["y2"][l], nhwc(taps[f"phi_subnet_{l}/y2"])); put_bn(net.phi[l]["bn2"], taps[f"phi_subnet_{l}/y2"])    # (level 2: the concat slice itself)
    put(b["concat"][..., 128:], nhwc(taps["concat"])[..., 128:])
    put(b["final"], nhwc(taps["final"])); put_bn(net.final_bn, taps["final"])
    put(b["logits"], heat.detach())
    losses = net.compute_losses({k: torch.tensor(v).cuda() for k, v in lab.items()})
    net.backward()
    got = {k: net.grads[k].cpu().numpy().astype(np.float64) for k in want}
    rel = lambda a, c: float(np.linalg.norm(a - c) / (np.linalg.norm(c) + 1e-300))
    cos = lambda a, c: float(a.ravel() @ c.ravel() / (np.linalg.norm(a) * np.linalg.norm(c) + 1e-300))
    rows = [(k, want[k].size, float(np.linalg.norm(want[k])), rel(got[k], want[k]), cos(got[k], want[k])) for k in sorted(want)]
    cat = lambda g: np.concatenate([g[k].ravel() for k in sorted(want)])
    out = {"rows": rows, "all_rel": rel(cat(got), cat(want)), "all_cos": cos(cat(got), cat(want)), "loss": float(losses[6]),
           "oracle_loss": float(total.detach()), "train_first": first, "train_last": last}
    if verbose:
        print(f"variables after {steps} steps (total loss {first} -> {last}); batch of {B} @ {size}^2; teacher-forced backward of the "
              f"{'bf16' if dtype == torch.bfloat16 else 'f32'} build vs the {'emulating ' if em else ''}oracle: total loss {out['loss']:.5f} vs "
              f"{out['oracle_loss']:.5f}; ALL gradients rel-L2 {out['all_rel']:.4f}, cosine {out['all_cos']:.6f}")
        print("%-62s %9s %10s %8s %9s" % ("tensor", "size", "|g|", "rel-L2", "cosine"))
        for r in rows:
            print("%-62s %9d %10.3e %8.4f %9.6f" % r)
        worst = sorted(rows, key=lambda r: -r[3])[:5]
        print("worst five:", [(r[0], round(r[3], 4)) for r in worst])
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    run(*a)
