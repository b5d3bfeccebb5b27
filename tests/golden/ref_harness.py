"""Import the *real* reference (/root/reference) in this container to make golden vectors.

Runs ONLY in the build container: /root/reference does not exist on the GPU box, and no
test imports this module.  What travels is data (``tests/golden/*.npz``) plus this script.

The reference needs packages this image lacks (pytorch_lightning, omegaconf, torchaudio,
torchvision, timm, av, ... and the un-vendored ``descript-audio-codec``).  None of them
takes part in the arithmetic of the hot path, so they are replaced by *import-only*
placeholders: modules whose attributes are inert objects.  The
arithmetic that runs is the reference's own: ``VAURAModel.generate`` /
``_sample_next_token`` (models/vaura_model.py:410-597, 775-827), ``Transformer``
(models/modules/sampler/llama.py), ``Pattern`` (models/modules/misc/codebook_patterns.py)
and ``sample_top_k/top_p/multinomial`` (utils/utils.py:139-196).
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"
_MISSING_ROOTS = (
    "pytorch_lightning", "omegaconf", "torchaudio", "torchvision", "timm", "av", "pyloudnorm",
    "decord", "pytorchvideo", "ffmpeg", "julius", "audiotools", "tensorboard", "torchmetrics",
    "lightning", "lightning_fabric", "cv2", "librosa", "soundfile", "matplotlib", "PIL", "fvcore",
    "iopath", "simplejson", "wandb",
)


class _Inert:
    """Callable, attribute-chaining placeholder usable as a base class."""

    def __init__(self, name="inert"):
        self.__dict__["_n"] = name

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Inert(f"{self._n}.{item}")

    def __call__(self, *a, **k):
        # decorator use: @x.y  /  @x.y(...)
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Inert(self._n + "()")

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())

    def __getitem__(self, item):
        return _Inert(self._n + "[]")


class _PlaceholderModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Inert(f"{self.__name__}.{item}")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _MISSING_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _PlaceholderModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _LightningModuleShim(nn.Module):
    """Only what ``VAURAModel.__init__`` / ``generate`` touch on the Lightning base."""

    def save_hyperparameters(self, *a, **k):
        pass

    @property
    def device(self):
        return next(self.parameters()).device

    def print(self, *a, **k):
        print(*a, **k)

    def log(self, *a, **k):
        pass


def _install_dac_placeholder():
    """``llama.py:20-21`` imports two names from descript-audio-codec: the ``DAC`` class (a type
    annotation) and ``WNConv1d`` (= weight-normed ``nn.Conv1d``, used by ``initialize_embeddings``)."""
    dac = types.ModuleType("dac"); dac.__path__ = []
    dac_model = types.ModuleType("dac.model")
    dac_nn = types.ModuleType("dac.nn"); dac_nn.__path__ = []
    dac_layers = types.ModuleType("dac.nn.layers")

    class DAC(nn.Module):
        pass

    def WNConv1d(*a, **k):
        return torch.nn.utils.weight_norm(nn.Conv1d(*a, **k))

    dac_model.DAC = DAC
    dac.DAC = DAC
    dac_layers.WNConv1d = WNConv1d
    dac.model, dac.nn, dac_nn.layers = dac_model, dac_nn, dac_layers
    sys.modules.update({"dac": dac, "dac.model": dac_model, "dac.nn": dac_nn, "dac.nn.layers": dac_layers})


def _install_avclip_placeholders():
    """What the Segment-AVCLIP extractor (models/modules/feature_extractors/avclip/) imports from packages this image lacks.
    None of it takes part in the eval-mode arithmetic: ``trunc_normal_`` only initialises parameters (the goldens load seeded
    weights over them), ``DropPath`` is the identity in eval mode (the standard stochastic-depth module, restated below), and
    ``to_2tuple`` turns 16 into (16, 16).  ``OmegaConf.load`` only has to read the backbone YAML into an object with attribute
    access and assignment (motionformer.py:124-137)."""
    import itertools
    import yaml

    class DropPath(nn.Module):
        def __init__(self, drop_prob: float = 0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else tuple(itertools.repeat(x, 2))

    layers = types.ModuleType("timm.models.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.DropPath = DropPath
    layers.to_2tuple = to_2tuple
    timm = _PlaceholderModule("timm"); timm.__path__ = []
    models = _PlaceholderModule("timm.models"); models.__path__ = []
    timm.models, models.layers = models, layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})

    class _Node(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    def _wrap(o):
        if isinstance(o, dict):
            return _Node({k: _wrap(v) for k, v in o.items()})
        return [_wrap(v) for v in o] if isinstance(o, list) else o

    class OmegaConf:
        @staticmethod
        def load(path):
            with open(path) as f:
                return _wrap(yaml.safe_load(f))

    import omegaconf  # placeholder module
    omegaconf.OmegaConf = OmegaConf


def build_reference_motionformer(sd: dict):
    """The reference's own ``MotionFormer`` in the configuration of configs/modules/feature_extractors/avclip_vggsound.yaml
    (no checkpoint file: the class then builds the 'divided_224_16x4' backbone, motionformer.py:112-114) with a seeded
    state dict loaded over its parameters."""
    install()
    _install_avclip_placeholders()
    import models.modules.feature_extractors.avclip  # noqa: F401  (appends the avclip dir to sys.path)
    from models.modules.feature_extractors.avclip.motionformer import MotionFormer as RefMotionFormer
    m = RefMotionFormer(extract_features=True, ckpt_path=None, factorize_space_time=True,
                        agg_space_module="TransformerEncoderLayer", agg_time_module="torch.nn.Identity", add_global_repr=False)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("patch_embed.proj.") for k in missing), missing      # the 2-D patch embedding is never used (:246-248)
    return m.eval()


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("the reference is only available in the build container")
    sys.meta_path.append(_Finder())
    import pytorch_lightning as pl  # placeholder
    pl.LightningModule = _LightningModuleShim
    pl.utilities = sys.modules.get("pytorch_lightning.utilities") or importlib.import_module("pytorch_lightning.utilities")
    pl.utilities.rank_zero_only = lambda f: f
    _install_dac_placeholder()
    sys.path.insert(0, REFERENCE_ROOT)
    os.chdir(REFERENCE_ROOT)  # some reference imports are cwd-relative
    _installed = True


# ------------------------------------------------------------------ stand-in plugins
class MotionFormer(nn.Module):
    """Pass-through feature extractor: the class *name* gates the AVCLIP branch
    (models/vaura_model.py:73-76); ``forward`` returns (feats (B,S,t,768), None)."""

    def __init__(self, **kw):
        super().__init__()
        self._p = nn.Parameter(torch.zeros(1), requires_grad=False)

    def forward(self, x):
        return x, None


class _Quantizer(nn.Module):
    def __init__(self, size, dim, latent):
        super().__init__()
        self.codebook = nn.Embedding(size, dim)
        self.out_proj = torch.nn.utils.weight_norm(nn.Conv1d(dim, latent, kernel_size=1))


class _FakeDac(nn.Module):
    def __init__(self, n_q=9, size=1024, dim=8, latent=1024):
        super().__init__()
        self.quantizer = nn.Module()
        self.quantizer.quantizers = nn.ModuleList(_Quantizer(size, dim, latent) for _ in range(n_q))
        self.sample_rate = 44100


class DacModelWrapper(nn.Module):
    """Shape-only codec stand-in: the real ``dac`` package is absent, and the codec's
    arithmetic is pinned separately (oracle/dac_oracle.py, 'parity unpinned' by the reference)."""

    decode_fn = None  # optionally set to the CPU DAC restatement

    def __init__(self, model_sr: int = 44100, ckpt_path=None, latent: int = 1024):
        super().__init__()
        self.model_sr = model_sr
        self.model = _FakeDac(latent=latent)

    def decode(self, codes):
        if type(codes) == list:
            codes = codes[0][0]
        if DacModelWrapper.decode_fn is not None:
            return DacModelWrapper.decode_fn(codes)
        return torch.zeros(codes.shape[0], 1, codes.shape[-1] * 512)

    @property
    def sample_rate(self):
        return self.model_sr


def build_reference_model(sampler_params: dict, sampler_sd: dict, n_q: int = 9):
    """Construct the reference ``VAURAModel`` with the real sampler/pattern classes and load
    a synthetic sampler state dict into it."""
    install()
    from models.vaura_model import VAURAModel  # noqa: E402  (reference)

    me = __name__
    model = VAURAModel(
        feature_extractor_config={"target": f"{me}.MotionFormer"},
        audio_encoder_config={"target": f"{me}.DacModelWrapper", "params": {"model_sr": 44100}},
        sampler_config={"target": "models.modules.sampler.llama.Transformer", "params": dict(sampler_params)},
        visual_bridge_config={"target": "torch.nn.Identity"},
        pattern_provider_config={
            "target": "models.modules.misc.codebook_patterns.DelayedPatternProvider",
            "params": {"n_q": n_q},
        },
        flatten_vis_feats=True,
        freeze_feature_extractor=True,
    )
    model.eval()
    model.sampler.audio_tokens_per_video_frame = 7  # scripts/generate.py:216
    missing, unexpected = model.sampler.load_state_dict(sampler_sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if "freqs" not in m], missing
    return model
