"""Helpers shared by the -m gpu parity tests (tests/ is on sys.path under pytest's rootdir conftest)."""
import torch


def assert_cfg_tokens_or_recorded_near_tie(tok, ref, margins, tol, what, threshold_gap=None):
    """Tokens must equal the reference's.  The one admissible exception: the FIRST differing token of a clip sits on a step whose
    recorded reference margin — CFG-mixed top-1 / top-2 logit gap (greedy) or the relative gap of argmax(p / Exp(1)) over the kept
    set (sampled; optionally the relative gap at the top-k threshold) — is below `tol`: there the 22-bit operand format's logit
    error (<= 1.2e-5, times up to 2 * cfg_scale - 1 = 11 through the CFG mix) can legitimately decide the draw, and every later
    token of that clip then differs.  Clips are independent, so each is judged on its own.  Returns clips that are identical."""
    K, T = tok.shape[1], tok.shape[2]
    steps = torch.arange(T)[None, :] + 1 + torch.arange(K)[:, None]                     # step that produced (k, t)
    same = 0
    for b in range(tok.shape[0]):
        if torch.equal(tok[b], ref[b]):
            same += 1
            continue
        bad = tok[b] != ref[b]
        s = int(steps[bad].min())
        for k in range(K):
            if bool((bad & (steps == s))[k].any()):
                m = float(margins[s - 1, b, k])
                g = float(threshold_gap[s - 1, b, k]) if threshold_gap is not None else float("inf")
                assert m < tol or g < tol, (f"{what}: clip {b} first differs at step {s}, codebook {k}: recorded reference margin {m:.3e}"
                                            f" (threshold gap {g:.3e}) >= {tol}")
        assert torch.equal(tok[b][steps < s], ref[b][steps < s])
        print(f"{what}: clip {b} identical up to step {s} (recorded near-tie there)")
    return same
