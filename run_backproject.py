#!/usr/bin/env python3
"""Command-line counterpart of `python backproject.py` (reference main(): backproject.py:301-336), on the fused
HIP path.  Flag names follow the reference's tyro flags; the 2-D feature network (LSeg / DINOv2 weights are not
available offline) is replaced by per-view feature maps read from --feature-maps, or by --synthetic inputs.

    python run_backproject.py --synthetic C1 --results-dir /tmp/out
    python run_backproject.py --data-dir data/garden --checkpoint ckpt.pt --format gsplat --data-factor 4 \
        --feature-maps feats/ --feature lseg --results-dir results/garden
    torchrun --nproc-per-node 8 run_backproject.py ...      # views shard over ranks, one RCCL all-reduce
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default="./data/garden")
    ap.add_argument("--checkpoint", default="./data/garden/ckpts/ckpt_29999_rank0.pt")
    ap.add_argument("--results-dir", default="./results/garden")
    ap.add_argument("--format", choices=["inria", "gsplat", "ply"], default="gsplat")
    ap.add_argument("--rasterizer", choices=["inria", "gsplat"], default=None, help="deprecated alias of --format")
    ap.add_argument("--data-factor", type=int, default=4)
    ap.add_argument("--feature", choices=["lseg", "dino"], default="lseg")
    ap.add_argument("--feature-maps", default=None, help="directory with <image name>.pt tensors [H,W,D]")
    ap.add_argument("--encoder", default=None, help="[512,16] encoder tensor (.pt): backproject_compressed.py")
    ap.add_argument("--synthetic", default=None, help="run a seeded synthetic config (C1, C2, ...) instead of files")
    ap.add_argument("--no-prune", action="store_true", help="skip the d > 0 pruning report (utils.prune_by_gradients)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise RuntimeError("a HIP device is required (the reference likewise requires CUDA, backproject.py:314)")
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    dev = torch.device("cuda", torch.cuda.current_device())
    rank = dist.get_rank() if dist.is_initialized() else 0

    import gsbp_amd
    from gsbp_amd import scene_io, synthetic as syn

    if args.synthetic:
        cfg = syn.CONFIGS[args.synthetic]
        means, quats, scales, opac = [t.to(dev) for t in syn.activate(syn.make_scene(cfg))]
        K, viewmats, W, H, dim = syn.intrinsics(cfg), syn.make_cameras(cfg), cfg.width, cfg.height, cfg.feat_dim
        encoder = syn.make_encoder(cfg).to(dev) if cfg.encoder_dim else None

        def feature_fn(v):
            return syn.make_feature_map(cfg, v, device=dev)
    else:
        splats = scene_io.load_checkpoint(args.checkpoint, args.data_dir, format=args.format,
                                          data_factor=args.data_factor, rasterizer=args.rasterizer)
        means, quats = splats["means"].to(dev).float(), splats["rotation"].to(dev).float()
        scales, opac = torch.exp(splats["scaling"]).to(dev).float(), torch.sigmoid(splats["opacity"]).to(dev).float()
        K = splats["camera_matrix"]
        W, H = int(K[0, 2] * 2), int(K[1, 2] * 2)  # backproject.py:85-86
        images = sorted(splats["colmap_project"].images.values(), key=lambda im: im.name)  # backproject.py:74
        viewmats = torch.stack([scene_io.get_viewmat_from_colmap_image(im) for im in images])
        if not args.feature_maps:
            raise SystemExit("--feature-maps is required (no LSeg/DINO weights offline)")
        encoder = torch.load(args.encoder).to(dev).float() if args.encoder else None
        first = torch.load(os.path.join(args.feature_maps, images[0].name + ".pt"))
        dim = first.shape[-1]

        def feature_fn(v):
            f = torch.load(os.path.join(args.feature_maps, images[v].name + ".pt")).to(dev).float()
            return f if f.shape[:2] == (H, W) else torch.nn.functional.interpolate(
                f.permute(2, 0, 1)[None], size=(H, W), mode="bilinear")[0].permute(1, 2, 0)

    reduction = "mean" if args.feature == "dino" else "sum"  # backproject.py:263,283 vs :127,145
    out, F, d, stats = gsbp_amd.create_feature_field(means, quats, scales, opac, viewmats, K, W, H, feature_fn, dim,
                                                     reduction=reduction, encoder=encoder, return_partials=True,
                                                     verbose=True)
    if rank == 0:
        if not args.no_prune:
            keep = gsbp_amd.prune_mask(d)
            print("Total splats", keep.numel())  # utils.py:258-260
            print("Pruned", int((~keep).sum()), "splats")
            print("Remaining", int(keep.sum()), "splats")
        name = "features_lseg_compressed.pt" if encoder is not None else f"features_{args.feature}.pt"
        print("saved", scene_io.save_features(out.cpu(), args.results_dir, name), tuple(out.shape), stats)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
