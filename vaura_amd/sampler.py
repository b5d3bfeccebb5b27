"""Drop-in sampler plugin: ``target: vaura_amd.sampler.Transformer``.

Host-side mirror of the reference's ``models.modules.sampler.llama.Transformer``
(/root/reference/models/modules/sampler/llama.py:286-586) for the generation path:
same constructor keywords (configs/modules/samplers/llama_9cbs.yaml:3-17), same state-dict
keys, same attributes the hosts read or write, same call
signature — but no arithmetic of its own: everything is executed by libvaura_hip.so through
``DecoderEngine``.  The class name must stay ``Transformer``: the host enables CFG only for
that name (models/vaura_model.py:786-788).

Training (``targets``/loss, dropout, drop-path) is out of scope; the module is inference-only.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .engine import DecoderEngine
from .synth import SamplerCfg


def _tree(root: nn.Module, shapes: Dict[str, tuple], buffers=()):
    """Create nested holder modules so that ``root.state_dict()`` has exactly the dotted keys given."""
    for key, shape in shapes.items():
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        t = torch.zeros(*shape)
        if key in buffers:
            mod.register_buffer(parts[-1], t)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
    _listify(root)


def _listify(mod: nn.Module):
    """Holders whose children are named 0..n-1 become ``nn.ModuleList`` (iterable, indexable;
    the state-dict keys are unchanged)."""
    for name, child in list(mod.named_children()):
        _listify(child)
        names = [n for n, _ in child.named_children()]
        if names and not list(child.named_parameters(recurse=False)) and names == [str(i) for i in range(len(names))]:
            setattr(mod, name, nn.ModuleList([getattr(child, n) for n in names]))


class Transformer(nn.Module):
    def __init__(self, num_layers: int = 12, d_model: int = 512, d_codebook: int = 1024, block_size_audio: int = 512,
                 block_size_video: int = 64, nhead: int = 8, dim_feedforward: int = 2048, dropout: float = 0.1,
                 activation: str = "relu", layer_norm_eps: float = 1e-5, batch_first: bool = False,
                 norm_first: bool = False, num_codebooks: int = 2, positional_embedder: str = "sinusoidal",
                 use_visual_conditioning: bool = True, use_delay_strategy: bool = False,
                 cond_feature_channel_scaler: int = 2, weight_dtype: str = "auto", one_launch_mlp: bool = True,
                 plane_shift: int = 0, near_tie: str = "report", kv_dtype: str = "f32"):
        super().__init__()
        layer_norm_eps = float(layer_norm_eps)      # PyYAML reads the reference's `layer_norm_eps: 1e-5` as a string (OmegaConf does not)
        self.cfg = SamplerCfg(num_layers=num_layers, d_model=d_model, nhead=nhead, d_codebook=d_codebook,
                              num_codebooks=num_codebooks, block_size_audio=block_size_audio,
                              block_size_video=block_size_video,
                              cond_feature_channel_scaler=cond_feature_channel_scaler, layer_norm_eps=layer_norm_eps)
        c = self.cfg
        # attributes the hosts touch
        self.num_codebooks = num_codebooks
        self.d_codebook = d_codebook
        self.vocab_size = d_codebook
        self.n_layer = num_layers
        self.block_size = c.block_size
        self.config = SimpleNamespace(block_size=c.block_size, dim=d_model, n_layer=num_layers, n_head=nhead,
                                      norm_eps=layer_norm_eps, rope_base=c.rope_base, initializer_range=0.02)
        self.audio_tokens_per_video_frame: Optional[int] = None
        self.codebook_pattern = None
        # Storage of the streamed matrices on the device (engine.resolve_weight_dtype).  "auto" (default): one fp16 plane when
        # that is lossless for the loaded checkpoint, else two ("h2": 22 significand bits).  "h1" / "fp8"
        # force a smaller storage and ROUND a checkpoint that does not fit it (not token-exact any more); "f32" forces fp32.
        self.weight_dtype = weight_dtype
        # w1||w3 -> w2 of a layer as one launch where eligible (engine.DecoderEngine): False when several processes share the GPU
        self.one_launch_mlp = one_launch_mlp
        # every activation plane set stored x 2^-plane_shift (engine.DecoderEngine): for a checkpoint known to carry massive activations
        self.plane_shift = int(plane_shift)
        # near-tie detector policy (engine.DecoderEngine: off | report | rerun) and K/V cache type (f32 | f16: configs[4] with "fp8h")
        self.near_tie = near_tie
        self.kv_dtype = kv_dtype

        D, F = c.d_model, c.ffn_dim
        shapes = {
            "empty_video_emb": (1, 1, c.cond_dim),
            "cls_embeddings.uncond_embedding": (c.uncond_tokens, c.cond_in),
            "cls_embeddings.projection.fc1.weight": (c.cond_dim, c.cond_in),
            "cls_embeddings.projection.fc2.weight": (c.cond_dim, c.cond_dim),
            "norm.weight": (D,),
        }
        for l in range(num_layers):
            p = f"layers.{l}."
            shapes.update({p + "attention.wqkv.weight": (3 * D, D), p + "attention.wo.weight": (D, D),
                           p + "feed_forward.w1.weight": (F, D), p + "feed_forward.w3.weight": (F, D),
                           p + "feed_forward.w2.weight": (D, F), p + "attention_norm.weight": (D,),
                           p + "ffn_norm.weight": (D,)})
        for k in range(num_codebooks):
            shapes[f"lm_heads.{k}.weight"] = (d_codebook, D)
            p = f"tok_embeddings.{k}."
            shapes.update({p + "emb.weight": (d_codebook + 1, c.codebook_dim),
                           p + "out_proj.weight_g": (c.tok_dim, 1, 1),
                           p + "out_proj.weight_v": (c.tok_dim, c.codebook_dim, 1),
                           p + "out_proj.bias": (c.tok_dim,)})
        _tree(self, shapes, buffers=("cls_embeddings.uncond_embedding",))
        self._engine: Optional[DecoderEngine] = None
        self._engine_key = None

    # ------------------------------------------------------------------ host hooks
    def initialize_embeddings(self, dac_model) -> None:
        """Token embeddings start from the codec's codebooks and output projections
        (llama.py:387-412): one extra row for the special token (id == d_codebook)."""
        for k, q in enumerate(dac_model.quantizer.quantizers):
            cb = q.codebook.weight.detach().float()
            extra = torch.randn(1, cb.shape[1]) * self.config.initializer_range
            emb = self.tok_embeddings[k].emb
            emb.weight.data = torch.cat([cb.cpu(), extra], dim=0).to(emb.weight.device)
            proj = self.tok_embeddings[k].out_proj
            proj.weight_g.data = q.out_proj.weight_g.detach().float().clone().to(proj.weight_g.device)
            proj.weight_v.data = q.out_proj.weight_v.detach().float().clone().to(proj.weight_v.device)
            proj.bias.data = q.out_proj.bias.detach().float().clone().to(proj.bias.device)
        self._engine = None

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        self._engine = None
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _weights_fingerprint(self):
        """Changes whenever a parameter is rewritten in place (``_version``), replaced (``data_ptr``) or moved: covers a
        ``load_state_dict`` on the PARENT module (which never calls this class's override), ``initialize_embeddings`` and
        manual edits, so a stale packed copy is never decoded from."""
        return tuple((t._version, t.data_ptr()) for t in list(self.parameters()) + list(self.buffers()))

    def engine(self) -> DecoderEngine:
        """Pack the current parameters for the HIP path (once per device / weight dtype / parameter state)."""
        dev = self.norm.weight.device
        key = (str(dev), self.weight_dtype, self.one_launch_mlp, self.plane_shift, self.near_tie, self.kv_dtype, self._weights_fingerprint())
        if self._engine is None or self._engine_key != key:
            if dev.type != "cuda":
                raise L.VauraHipError("vaura_amd.sampler.Transformer runs on a HIP device only; call .to('cuda') first")
            sd = {k: v for k, v in self.state_dict().items()}
            self._engine = DecoderEngine(self.cfg, sd, dev, wdtype=self.weight_dtype, one_launch_mlp=self.one_launch_mlp,
                                         plane_shift=self.plane_shift if self.weight_dtype != "f32" else 0,
                                         near_tie=self.near_tie, kv_dtype=self.kv_dtype)
            self._engine_key = key
        return self._engine

    @property
    def resolved_weight_dtype(self) -> str:
        """What "auto" resolved to for the parameters currently loaded ("h1" | "h2" | "fp8" | "f32")."""
        return self.engine().wdtype

    # ------------------------------------------------------------------ reference call surface
    @torch.no_grad()
    def forward(self, tgt: torch.Tensor, memory: torch.Tensor, use_conditioning: bool = True, tgt_mask=None,
                memory_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None, tgt_is_causal: bool = False,
                memory_is_causal: bool = False, return_attention_weights: bool = False,
                apply_per_video_frame_mask: bool = False):
        """tgt (Bs, K, L) int64, memory (Bs, Tv, 768) -> (logits (Bs, K, L, d_codebook), None, None)
        — llama.py:520-539.  The extra keywords are accepted and ignored, as in the reference."""
        if self.audio_tokens_per_video_frame is None:
            raise L.VauraHipError("audio_tokens_per_video_frame must be set (scripts/generate.py:216 sets 7)")
        # the reference host re-feeds the whole prefix every step (vaura_model.py:504-506); positions already seen with the same
        # condition are served from the engine's K/V + logits cache (DecoderEngine.forward_cached)
        logits = self.engine().forward_cached(tgt, memory.float(), self.audio_tokens_per_video_frame)
        return logits, None, None
