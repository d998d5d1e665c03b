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
    # The reference's main() also takes these two (backproject.py:309-310).  There feature_field_batch_count is handed to
    # create_feature_field_lseg and never read (backproject.py:25-27,77-82), and run_feature_field_on_cpu only moves the LSeg
    # NETWORK to the CPU (backproject.py:30-41) -- the rasterisation stays on the GPU either way.  Here the 2-D network is
    # replaced by supplied feature maps, so both are accepted for command-line compatibility and change nothing.
    ap.add_argument("--feature-field-batch-count", type=int, default=1,
                    help="accepted for compatibility with the reference's main() (unused there as well)")
    ap.add_argument("--run-feature-field-on-cpu", action=argparse.BooleanOptionalAction, default=False,
                    help="accepted for compatibility: the reference moves only its LSeg network to the CPU with it; the "
                         "feature maps are supplied here, the back-projection always runs on the GPU")
    ap.add_argument("--feature-maps", default=None, help="directory with <image name>.pt tensors [H,W,D]")
    ap.add_argument("--encoder", default=None, help="[512,16] encoder tensor (.pt): backproject_compressed.py")
    ap.add_argument("--synthetic", default=None, help="run a seeded synthetic config (C1, C2, ...) instead of files")
    ap.add_argument("--no-prune", action="store_true",
                    help="skip the pruning step and build the field on ALL Gaussians (the default, like the reference's main(), "
                         "is prune_by_gradients -> test_proper_pruning -> build on the pruned scene, backproject.py:320-325)")
    ap.add_argument("--prune-by-product", action="store_true",
                    help="ONE sweep instead of two: build the field on all Gaussians and take the mask from the same "
                         "denominators (keep = d > 0), then drop the pruned rows.  Not the reference's arithmetic to the last "
                         "digit: a pruned Gaussian has no weight anywhere but may be the one that TERMINATES pixels "
                         "(T' <= 1e-4), so building with it present moves some kept rows (C2 size, two views: median 7e-9, "
                         "99 %% of the rows within 1.2e-7, 0.4 %% beyond 1e-3 of the prune-first result; tests/test_gpu_pruning.py)")
    ap.add_argument("--dist-backend", default="nccl", help="process-group backend under torchrun (nccl = RCCL over xGMI)")
    ap.add_argument("--one-device", action="store_true",
                    help="every rank uses cuda:0 (with --dist-backend gloo: the N > 1 bookkeeping on a one-GPU box)")
    args = ap.parse_args()

    import gsbp_amd  # BEFORE the first HIP call: the package asks the runtime for the hardware queues its view pipeline needs
    if not torch.cuda.is_available():
        raise RuntimeError("a HIP device is required (the reference likewise requires CUDA, backproject.py:314)")
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(args.dist_backend)
    dev = torch.device("cuda", torch.cuda.current_device())
    rank = dist.get_rank() if dist.is_initialized() else 0

    from gsbp_amd import scene_io, synthetic as syn

    if args.synthetic:
        cfg = syn.CONFIGS[args.synthetic]
        splats = {k: v.to(dev) for k, v in syn.make_scene(cfg).items()}  # pre-activation, the reference's key names
        means, quats, scales, opac = syn.activate(splats)
        K, viewmats, W, H, dim = syn.intrinsics(cfg), syn.make_cameras(cfg), cfg.width, cfg.height, cfg.feat_dim
        encoder = syn.make_encoder(cfg).to(dev) if cfg.encoder_dim else None

        # (DINO64 / LSEG480: the network's own low-resolution map, upsampled inside the kernels like the reference's F.interpolate)
        upsample = cfg.upsample
        reduction = cfg.reduction if cfg.lowres else ("mean" if args.feature == "dino" else "sum")

        def feature_fn(v):
            return syn.make_feature_map(cfg, v, device=dev)
    else:
        splats = scene_io.load_checkpoint(args.checkpoint, args.data_dir, format=args.format,
                                          data_factor=args.data_factor, rasterizer=args.rasterizer)
        means, quats = splats["means"].to(dev).float(), splats["rotation"].to(dev).float()
        scales, opac = torch.exp(splats["scaling"]).to(dev).float(), torch.sigmoid(splats["opacity"]).to(dev).float()
        splats = {k: (v.to(dev).float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in splats.items()}
        K = splats["camera_matrix"]
        W, H = int(K[0, 2] * 2), int(K[1, 2] * 2)  # backproject.py:85-86
        images = sorted(splats["colmap_project"].images.values(), key=lambda im: im.name)  # backproject.py:74
        viewmats = torch.stack([scene_io.get_viewmat_from_colmap_image(im) for im in images])
        if not args.feature_maps:
            raise SystemExit("--feature-maps is required (no LSeg/DINO weights offline)")
        encoder = torch.load(args.encoder).to(dev).float() if args.encoder else None
        first = torch.load(os.path.join(args.feature_maps, images[0].name + ".pt"))
        dim = first.shape[-1]
        # A map at the network's resolution is upsampled the way the reference does it -- bilinear for lseg (backproject.py:110-112),
        # nearest for dino's patch tokens (:244-248) -- INSIDE the kernels (dino maps whose tokens cover a tile: token space); with an
        # encoder the map is materialised first (the encoder-fused kernels read full-resolution pixels)
        mode = "nearest" if args.feature == "dino" else "bilinear"
        upsample = mode if (tuple(first.shape[:2]) != (H, W) and encoder is None) else None
        reduction = "mean" if args.feature == "dino" else "sum"  # backproject.py:263,283 vs :127,145

        def feature_fn(v):
            f = torch.load(os.path.join(args.feature_maps, images[v].name + ".pt")).to(dev).float()
            if upsample is not None or tuple(f.shape[:2]) == (H, W):
                return f
            kw = {"align_corners": False} if mode == "bilinear" else {}
            return torch.nn.functional.interpolate(f.permute(2, 0, 1)[None], size=(H, W), mode=mode, **kw)[0].permute(1, 2, 0)

    # backproject.py:323-325: splats_optimized = prune_by_gradients(splats); test_proper_pruning(splats, splats_optimized);
    # the field is then built on the PRUNED scene.  The mask costs one blend per view (no scatter).  Under a process group
    # the sweep AND the render check are sharded over the ranks by view like the field build itself (round 4 had rank 0 do
    # both alone while the others waited: ~0.3 s at C2 against a 0.09 s field build per rank at 8 GPUs); the weight sums are
    # all-reduced, so every rank holds the SAME mask and the shapes of the collectives that follow agree by construction.
    n_all = means.shape[0]
    keep = None
    vm_dev, K_dev = viewmats.to(dev), K.to(dev)

    def report_and_check(keep):
        if rank == 0:
            print("Total splats", keep.numel())  # utils.py:258-260
            print("Pruned", int((~keep).sum()), "splats")
            print("Remaining", int(keep.sum()), "splats")
        if "features_dc" in splats:  # utils.test_proper_pruning renders with the SH colours (checkpoints only)
            pruned = {k: (v[keep] if k in gsbp_amd.pruning._PER_GAUSSIAN else v) for k, v in splats.items()}
            rep = gsbp_amd.check_proper_pruning(splats, pruned, vm_dev, K_dev, W, H)
            if rank == 0:
                print("Percentage pruned: ", rep["percentage_pruned"])  # utils.py:348-359
                print("Max pixel error: ", rep["max_pixel_error"])
                print("Total pixel error: ", rep["total_pixel_error"])

    if not args.no_prune and not args.prune_by_product:
        keep = gsbp_amd.pruning.gradient_mask(splats, vm_dev, K_dev, W, H)
        report_and_check(keep)
        means, quats, scales, opac = means[keep], quats[keep], scales[keep], opac[keep]

    out, F, d, stats = gsbp_amd.create_feature_field(means, quats, scales, opac, viewmats, K, W, H, feature_fn, dim,
                                                     reduction=reduction, encoder=encoder, return_partials=True,
                                                     verbose=True, upsample=upsample)
    if args.prune_by_product and not args.no_prune:
        # "the mask comes free from the fused kernel" -- d is the all-reduced denominator of every Gaussian,
        # identical on every rank
        keep = d > 0
        report_and_check(keep)
        out = out[keep]
    if rank == 0:
        name = "features_lseg_compressed.pt" if encoder is not None else f"features_{args.feature}.pt"
        print("saved", scene_io.save_features(out.cpu(), args.results_dir, name), tuple(out.shape),
              f"(of {n_all} Gaussians)", stats)
        if keep is not None:  # the rows of the features file are the kept Gaussians, in order (like the reference's)
            torch.save(keep.cpu(), os.path.join(args.results_dir, "prune_mask.pt"))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
