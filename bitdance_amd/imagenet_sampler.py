"""FID-50k data-parallel sampler for the class-conditional ImageNet models on the native engine.

Mirrors ``/root/reference/imagenet_gen/sample_ddp.py`` / ``sample_ddp_parallel.py`` (identical scripts up to the model module;
SURVEY.md section 8f row 4): one process per GPU, rank r of W draws the classes of images ``W*n*i + r*n .. + n`` of the fixed
class list (:134-136), seeds its generator with ``seed * W + rank`` (:69-70), samples, converts to uint8 (:166-171), writes
``{index:06d}.png`` and rank 0 packs the folder into ``.npz`` (:29-61).  The path shards by image with no data-path collective
(the reference's only collectives are barriers): ranks are independent replicas, ``torch.distributed`` carries the barriers.

Launch (one node):  ``python -m torch.distributed.run --nproc-per-node N -m bitdance_amd.imagenet_sampler --model BitDance-B
--parallel-num 16 --ckpt BitDance_B_16x.pt --trained-vae ae_d16c32.pt --cfg-scale 6.1 --per-proc-batch-size 384 --to-npz``
(the reference's argument names; ``--parallel-num 1`` = the 1x checkpoints of ``sample_ddp.py``).

The index arithmetic lives in pure functions (``class_list``, ``rank_plan``, ``to_uint8``) so that it is tested without a GPU.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

MODELS = {  # imagenet_gen/src/model.py:394-430 == model_parallel.py:437-473
    "BitDance-B": dict(n_layer=24, n_head=12, dim=768, diff_layers=6, diff_dim=768, diff_adanln_layers=2),
    "BitDance-L": dict(n_layer=32, n_head=16, dim=1024, diff_layers=8, diff_dim=1024, diff_adanln_layers=2),
    "BitDance-H": dict(n_layer=40, n_head=20, dim=1280, diff_layers=12, diff_dim=1280, diff_adanln_layers=3),
}


def class_list(num_fid_samples: int, num_classes: int) -> np.ndarray:
    """sample_ddp_parallel.py:134-136: every class ``num_fid_samples // num_classes`` times in order, then zero padding."""
    per = num_fid_samples // num_classes
    return np.hstack([np.arange(0, num_classes).repeat(per), np.zeros(50000)])


def rank_plan(num_fid_samples: int, num_classes: int, per_proc_batch: int, world_size: int, rank: int):
    """The (first image index, class labels, images to keep) of every ``model.sample`` call this rank makes
    (sample_ddp_parallel.py:138-183): iteration i covers images ``world*n*i + rank*n .. + n``; a call is made while its first
    index is below ``num_fid_samples``; images at or beyond it are sampled but not saved."""
    labels = class_list(num_fid_samples, num_classes)
    n = per_proc_batch
    out = []
    for i in range(num_fid_samples // (n * world_size) + 1):
        start = world_size * n * i + rank * n
        if start >= num_fid_samples:
            break
        lab = labels[start:start + n]
        if len(lab) == 0:
            break
        out.append((start, lab.astype(np.int64), max(0, min(n, num_fid_samples - start))))
    return out


def rank_seed(seed: int, world_size: int, rank: int) -> int:
    return seed * world_size + rank                              # :69


def to_uint8(samples: torch.Tensor) -> np.ndarray:
    """[n, 3, H, W] in about [-1, 1] -> uint8 [n, H, W, 3] (:166-171)."""
    return torch.clamp(127.5 * samples + 128.0, 0, 255).permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()


def folder_name(args) -> str:
    ck = os.path.basename(args.ckpt).replace(".pth", "").replace(".pt", "")   # :102-108
    return (f"{args.model.replace('/', '-')}-{ck}-size-{args.image_size}-steps-{args.sample_steps}-cfg-{args.cfg_scale}-"
            f"seed-{args.seed}")


def pack_npz(sample_dir: str, num: int, compressed: bool = False, delete_folder: bool = True) -> str:
    """create_npz_from_sample_folder :29-61."""
    from PIL import Image
    first = np.asarray(Image.open(os.path.join(sample_dir, f"{0:06d}.png")).convert("RGB"), dtype=np.uint8)
    arr = np.empty((num, *first.shape), dtype=np.uint8)
    arr[0] = first
    for i in range(1, num):
        a = np.asarray(Image.open(os.path.join(sample_dir, f"{i:06d}.png")).convert("RGB"), dtype=np.uint8)
        if a.shape != first.shape:
            raise ValueError(f"Image shape mismatch at index {i}: got {a.shape}, expected {first.shape}")
        arr[i] = a
    path = f"{sample_dir}.npz"
    (np.savez_compressed if compressed else np.savez)(path, arr_0=arr)
    if delete_folder:
        import shutil
        shutil.rmtree(sample_dir)
    return path


def get_args(argv=None):
    p = argparse.ArgumentParser()                                # get_model_args (model_parallel.py:14-37) + the script's own
    p.add_argument("--model", type=str, choices=list(MODELS), default="BitDance-L")
    p.add_argument("--image-size", type=int, choices=[256, 512], default=256)
    p.add_argument("--down-size", type=int, default=16, choices=[16])
    p.add_argument("--patch-size", type=int, default=1, choices=[1, 2, 4])
    p.add_argument("--num-classes", type=int, default=1000)
    p.add_argument("--cls-token-num", type=int, default=64)
    p.add_argument("--latent-dim", type=int, default=16)
    p.add_argument("--parallel-num", type=int, default=1, choices=[1, 4, 16])
    p.add_argument("--time-shift", type=float, default=1.0)
    p.add_argument("--trained-vae", type=str, default="")
    p.add_argument("--ckpt", type=str, default=None)
    p.add_argument("--sample-dir", type=str, default="samples")
    p.add_argument("--per-proc-batch-size", type=int, default=32)
    p.add_argument("--num-fid-samples", type=int, default=50000)
    p.add_argument("--cfg-scale", type=float, default=4.6)
    p.add_argument("--seed", type=int, default=99)
    p.add_argument("--sample-steps", type=int, default=100)
    p.add_argument("--no-ema", action="store_true")
    p.add_argument("--mixed-precision", type=str, default="bf16", choices=["none", "bf16"])
    p.add_argument("--to-npz", action="store_true")
    p.add_argument("--chunk-size", type=int, default=0)
    return p.parse_args(argv)


def load_model(args, device):
    from .autoencoder import VQModel
    from .imagenet import BitDance
    ck = torch.load(args.ckpt, map_location="cpu", weights_only=False)
    if "ema" in ck and not args.no_ema:                          # :84-90
        sd = ck["ema"]
    elif "model" in ck:
        sd = ck["model"]
    else:
        raise Exception("please check model weight")
    ddconfig = None
    vae = None
    vae_keys = {k[len("vae."):]: v for k, v in sd.items() if k.startswith("vae.")}
    if not args.trained_vae and not vae_keys:
        # without a tokenizer BitDance.sample returns LATENTS [n, C, h, w]; the png / npz writer below would fail (or write
        # garbage) only after a whole batch has been sampled.  The reference builds its VAE unconditionally
        # (model_parallel.py:137-150) and loads vae.* strictly with the checkpoint.
        raise ValueError("no tokenizer weights: pass --trained-vae, or use a checkpoint that carries vae.* tensors")
    if args.trained_vae or vae_keys:
        ddconfig = dict(double_z=False, z_channels=args.latent_dim, in_channels=3, out_ch=3, ch=256, ch_mult=[1, 1, 2, 2, 4],
                        num_res_blocks=4)                        # model_parallel.py:137-146
        vae = VQModel(ddconfig=ddconfig, gan_decoder=False)
        if vae_keys:                                             # the checkpoint's own tokenizer first (the reference loads it strictly)
            vae.load_state_dict(vae_keys, strict=False)
        if args.trained_vae:                                     # then the override, as the reference does (:146-150)
            state = torch.load(args.trained_vae, map_location="cpu")
            vae.load_state_dict(state["state_dict"], strict=False)
        vae = vae.to(device).eval()
    return BitDance(sd, latent_dim=args.latent_dim, resolution=args.image_size, down_size=args.down_size,
                    patch_size=args.patch_size, cls_token_num=args.cls_token_num, num_classes=args.num_classes,
                    parallel_num=args.parallel_num, time_shift=args.time_shift, device=device, vae=vae, **MODELS[args.model])


def main(argv=None) -> int:
    import torch.distributed as dist
    args = get_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("the native sampler needs a GPU (HIP engine); there is no CPU fallback")
    if args.mixed_precision != "bf16":
        raise NotImplementedError("the native engine implements the bf16 autocast flow (--mixed-precision bf16)")
    dist.init_process_group("nccl")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    torch.manual_seed(rank_seed(args.seed, world, rank))
    torch.set_grad_enabled(False)
    model = load_model(args, f"cuda:{device}")
    out_dir = f"{args.sample_dir}/{folder_name(args)}"
    if os.path.isfile(out_dir + ".npz"):
        dist.barrier()
        dist.destroy_process_group()
        return 1
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    dist.barrier()
    from PIL import Image
    t0 = time.time()
    for it, (start, labels, keep) in enumerate(rank_plan(args.num_fid_samples, args.num_classes, args.per_proc_batch_size, world, rank)):
        # the reference samples under autocast(precision) (:157-163): here that governs the torch parts only (first step, VAE
        # decode); the native engine always computes in the bf16 flow, so --mixed-precision none is refused
        with torch.autocast("cuda", dtype=torch.bfloat16):
            imgs = model.sample(torch.from_numpy(labels).long().to(device), sample_steps=args.sample_steps,
                                cfg_scale=args.cfg_scale, chunk_size=args.chunk_size)
        for b, im in enumerate(to_uint8(imgs)[:keep]):
            Image.fromarray(im).save(f"{out_dir}/{start + b:06d}.png")
        if rank == 0 and it % 10 == 0:
            print(f"Step {it}, sampled so far (approx): {(it + 1) * len(labels) * world}, cost {time.time() - t0:.2f} s", flush=True)
    dist.barrier()
    if rank == 0 and args.to_npz:
        pack_npz(out_dir, args.num_fid_samples)
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
