"""One rank of a data-parallel UNFROZEN training step -- the worker tests/test_gpu_train_unfrozen.py starts twice (gloo on one
GPU; under RCCL on an 8-GPU node the same script is one rank per GPU: torchrun --nproc-per-node N tools/unfrozen_dp_worker.py --out DIR).
Every rank builds the same `small` policy, takes its slice of ONE fixed batch, runs FastVLAPolicy.fused_train_step with the gradient leaving per
bucket under the backward pass (BucketedGradExchange through fv_bucket_cb), and writes its reduced gradient + updated parameters to --out."""
import argparse
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "vla-from-fastvlm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def fixed_batch(B, dev):
    g = torch.Generator().manual_seed(6)
    return {"images": torch.rand(B, 3, 96, 128, generator=g).to(dev), "states": torch.randn(B, 14, generator=g).to(dev),
            "actions": torch.randn(B, 14, generator=g).to(dev), "tasks": ["pick up the red cube", "open the drawer", "push", "stack the blocks"][:B]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--min-numel", type=int, default=1 << 14)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    from vla_fastvlm.fastvla import FastVLAConfig, FastVLAPolicy
    torch.manual_seed(5)
    pol = FastVLAPolicy(FastVLAConfig(vlm_model_name="synthetic:small:41", hidden_dim=64, fusion_dim=64, dropout=0.0, freeze_backbone=False)).to(dev)
    pol.train()
    st = pol.enable_backbone_training(bucket_min_numel=args.min_numel)
    batch = fixed_batch(args.batch, dev)
    per = args.batch // world
    mine = {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
    out = pol.fused_train_step(mine, lr=1e-3, weight_decay=0.0)
    torch.cuda.synchronize()
    torch.save({"grads": st.g.cpu() / world / st.eng.train_loss_scale(), "flat": st.flat.cpu(), "loss": float(out["loss"]), "grad_norm": float(out["grad_norm"]),
                "collectives": list(st.bucketed.launched), "world": world}, Path(args.out) / f"rank{rank}.pt")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pol.model.backbone.engine().close()


if __name__ == "__main__":
    main()
