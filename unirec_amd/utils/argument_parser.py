"""Config defaults for the hot path, key-compatible with the reference's flat config dict
(unirec/config/base.yaml, config/model/{SASRec,GRU,MF}.yaml; merge order of unirec/utils/argument_parser.py:214-241:
base < model < dataset < caller's dict)."""
BASE = dict(seed=2022, init_method="normal", init_std=0.02, init_mean=0.0, scheduler="off", scheduler_factor=0.1,
            has_user_emb=False, has_user_bias=False, has_item_bias=False, use_position_emb=True, embedding_size=32,
            inner_size=128, dropout_prob=0.0, epochs=200, batch_size=400, learning_rate=0.001, optimizer="adam",
            early_stop=5, weight_decay=0.0, shuffle_train=False, loss_type="bce", distance_type="dot", max_seq_len=10,
            history_mask_mode="unorder", tau=1.0, n_sample_neg_train=4, use_tensorboard=False, use_wandb=False,
            train_file_format="user-item", key_metric="hit@5", metrics="['hit@1;5;10', 'ndcg@5;10', 'mrr', 'group_auc']",
            embedding_optimizer="lazy_dense", output_path="./output", verbose=1)
MODEL = {
    # the reference's config/model/*.yaml, value for value
    "SASRec": dict(n_layers=2, n_heads=16, inner_size=512, hidden_dropout_prob=0.5, attn_dropout_prob=0.5, hidden_act="swish",
                   layer_norm_eps=1e-10),
    "GRU": dict(embedding_size=64, max_seq_len=10, hidden_size=768),
    "MF": dict(embedding_size=64, has_user_emb=True),
    "ConvFormer": dict(n_layers=2, conv_size=50, inner_size=256, hidden_dropout_prob=0.5, padding_mode="circular", hidden_act="gelu",
                       layer_norm_eps=1e-9, seq_decay=-0.3, seq_merge=False, init_ratio=0.005),
    "FASTConvFormer": dict(n_layers=2, conv_size=50, inner_size=256, hidden_dropout_prob=0.5, padding_mode="constant", hidden_act="gelu",
                           layer_norm_eps=1e-9, seq_decay=-0.3, seq_merge=False, init_ratio=0.005),
    "AvgHist": dict(embedding_size=64, asymmetric=True, user_sequence_alpha=0.5, max_seq_len=10),
    "SVDPlusPlus": dict(embedding_size=64, user_sequence_alpha=0.5, max_seq_len=10, has_user_emb=True),
    "AttHist": dict(embedding_size=64, max_seq_len=10),
}


def parse_arguments(args: dict) -> dict:
    cfg = dict(BASE)
    cfg.update(MODEL.get(args.get("model", ""), {}))
    cfg.update({k: v for k, v in args.items() if v not in ("none", "None")})
    cfg.setdefault("exp_name", cfg.get("model", "unirec_amd"))
    return cfg
