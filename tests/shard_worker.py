"""Child program of tests/test_gpu_multirank.py: one rank of a clip-parallel job (vaura_amd/dist.py), the way
bench.py --gpus N runs it.  Launched by torch.distributed.run; on a 1-GPU box every rank uses cuda:0
(VAURA_BENCH_SHARE_GPU=1) and the collectives run over gloo on host copies (VAURA_BENCH_BACKEND=gloo) — the product
code path (sharding, clip_base-keyed noise, decode loop, codec, final gather of DEVICE tensors) is the same."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaura_amd import dist as vdist  # noqa: E402
from vaura_amd import synth  # noqa: E402
from vaura_amd.engine import CodecEngine, DecoderEngine  # noqa: E402


def run(total, layers, T, first, n, dev):
    cfg = synth.tiny_sampler(layers)
    eng = DecoderEngine(cfg, synth.sampler_state_dict(cfg, seed=21, round_bf16=True), dev, wdtype="bf16",
                        one_launch_mlp=os.environ.get("VAURA_BENCH_SHARE_GPU") != "1")      # ranks sharing ONE GPU: two launches
    ccfg = synth.FULL_CODEC
    codec = CodecEngine(ccfg, synth.codec_state_dict(ccfg, seed=22), dev)
    feats = synth.video_features(n, seed=23, first_clip=first).to(dev)
    codes = eng.generate_codes(feats, T, use_sampling=True, temp=1.0, top_k=250, cfg_scale=6.0, seed=77, clip_base=first)
    wav = codec.decode(codes)
    return codes, wav


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--total", type=int, default=8)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--frames", type=int, default=24)
    a = ap.parse_args()
    share = os.environ.get("VAURA_BENCH_SHARE_GPU") == "1"
    rank, local, world = vdist.init(os.environ.get("VAURA_BENCH_BACKEND", "nccl"), device_index=0 if share else None)
    if share:
        local = 0
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    first, n = vdist.shard(a.total, rank, world)
    counts = [vdist.shard(a.total, r, world)[1] for r in range(world)]
    codes, wav = run(a.total, a.layers, a.frames, first, n, dev)
    all_codes = vdist.gather_clips(codes.to(torch.int32), counts)      # device tensors through the gather
    all_wav = vdist.gather_clips(wav, counts)
    assert all_codes.is_cuda and all_wav.is_cuda
    vdist.barrier()
    seen = vdist.ranks_seen(dev)                                       # all_gather_object over the product backend
    worst = vdist.max_over_ranks(float(rank), dev)                     # all_reduce(MAX) on a device tensor
    if rank == 0:
        np.savez(a.out, codes=all_codes.cpu().numpy(), wav=all_wav.cpu().numpy(), world=world,
                 backend=(torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"), n_seen=len(seen), worst=worst)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
