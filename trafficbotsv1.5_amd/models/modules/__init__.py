from . import action_head, add_navi_latent, attention_rpe, distributions, input_encoder, mlp, polyline_encoder, transformer_rpe  # noqa: F401
