"""Which Motionformer backbones can the REFERENCE itself run?  (build container only: imports /root/reference)

    python tests/golden/check_reference_backbones.py

The reference picks the backbone config from the checkpoint (motionformer.py:82-114): divided_224_16x4.yaml,
motionformer_224_16x4.yaml (ATTN_LAYER: trajectory) or joint_224_16x4.yaml.  This script builds the reference's own
``VisionTransformer`` (video_model_builder.py) from each YAML, patched exactly like motionformer.py:133-138, at depth 1, and
calls ``forward_features`` the way ``MotionFormer.forward_segments`` does (motionformer.py:308).  Result in this tree:

  divided     runs  (DividedSpaceTimeBlock.forward accepts tok_mask, vit_helper.py:443-450)
  trajectory  TypeError: Block.forward() got an unexpected keyword argument 'tok_mask'
  joint       TypeError: the same call

because ``forward_features`` passes ``tok_mask=tok_mask`` to EVERY block (video_model_builder.py:266-268) and ``Block.forward``
(vit_helper.py:379-390) has no such parameter.  So a checkpoint that selects the trajectory or joint backbone cannot be run by
the reference at all; the divided backbone is the reference's only executable feature extractor, and the only one this build
provides (vaura_amd/feature_extractor.py).  Exit status 0 when the finding above still holds.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def run_backbone(name: str) -> str:
    import ref_harness as rh
    rh.install()
    rh._install_avclip_placeholders()
    import models.modules.feature_extractors.avclip  # noqa: F401
    import omegaconf
    from models.modules.feature_extractors.avclip.motionformer_src.video_model_builder import VisionTransformer
    cfg = omegaconf.OmegaConf.load(os.path.join(rh.REFERENCE_ROOT, "models/modules/feature_extractors/avclip/motionformer_src", name))
    cfg.VIT.ATTN_DROPOUT = 0.0                              # motionformer.py:133-138
    cfg.VIT.POS_EMBED = "joint" if name.startswith("joint") else "separate"
    cfg.VIT.USE_ORIGINAL_TRAJ_ATTN_CODE = True
    cfg.VIT.APPROX_ATTN_TYPE = "none"
    cfg.VIT.APPROX_ATTN_DIM = 64
    cfg.VIT.DEPTH = 1
    m = VisionTransformer(cfg).eval()
    x = torch.zeros(1, 1, 3, 16, 224, 224)
    try:
        with torch.no_grad():
            y, _ = m.forward_features(x)
        return f"runs {tuple(y.shape)}"
    except TypeError as e:
        return f"TypeError: {e}"


def main() -> int:
    res = {n: run_backbone(n) for n in ("divided_224_16x4.yaml", "motionformer_224_16x4.yaml", "joint_224_16x4.yaml")}
    for k, v in res.items():
        print(f"{k:28s} {v}")
    ok = (res["divided_224_16x4.yaml"].startswith("runs") and "tok_mask" in res["motionformer_224_16x4.yaml"]
          and "tok_mask" in res["joint_224_16x4.yaml"])
    print("finding holds: only the divided backbone is executable in the reference" if ok else "FINDING CHANGED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
