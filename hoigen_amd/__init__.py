"""hoigen_amd — MI355X (gfx950) native implementation of HOIGen's CLIP-encoder + CoOp-VAE hot path.

Drop-in surface (reference: soberguo/HOIGen):
    hoigen_amd.clip.load / tokenize / available_models      <- clipnet/clip.py
    hoigen_amd.model.build_model / CLIP / VisionTransformer  <- clipnet/model.py, CLIP_models_adapter_prior2.py
    hoigen_amd.vae.Encoder / Generator / PromptLearner_* / TextEncoder / mlp_net / vae_loss
                                                             <- main_coop_vae.py, finetune_ship.py
    hoigen_amd.distributed.encode_image_sharded              <- crops sharded over ranks + RCCL all-gather

All compute runs in hand-written HIP kernels behind the C ABI of ``include/hoigen_amd.h``
(``hoigen_amd/csrc``); there is no CPU or PyTorch fallback.
"""
from .clip import available_models, load, tokenize  # noqa: F401
from .model import CLIP, build_model  # noqa: F401

__version__ = "0.1.0"
