"""DCN with the reference's constructor (recman/tf/core/DCN.py:28-51).  The reference's
CrossNet layer does not exist (DCN.py:7); the default is the DCN-v1 vector form the README
cites (arXiv 1708.05123), fused over all layers in csrc/cross.hip; cross_type="matrix" (an
extra keyword, not in the reference) selects x_{l+1} = x0 o (W x_l + b) + x_l on the MFMA."""
from sklearn.metrics import log_loss, roc_auc_score

from .DeepModel import DeepModel


class DCN(DeepModel):
    model = "dcn"

    def __init__(self, feat_dict, embedding_size=8, embedding_l2_reg=0.00001, linear_l2_reg=0.00001,
                 deep_hidden_units=(32, 32), deep_dropout=(0.6, 0.6, 0.6), deep_activation="relu",
                 deep_l2_reg=0.0, cross_layer_num=3, cross_layer_l2_reg=0.0, epoch=10, batch_size=64,
                 learning_rate=0.001, optimizer="adam", random_seed=2019, use_linear=True,
                 loss_type="logloss", eval_metric=(roc_auc_score, log_loss), what_means_greater=None,
                 use_interactive_session=False, log_dir="./logs", strict_reference=False,
                 device="cuda", cross_type="vector"):
        assert loss_type in ["logloss", "mse"]
        hp = dict(embedding_size=embedding_size, embedding_l2_reg=embedding_l2_reg,
                  linear_l2_reg=linear_l2_reg, deep_hidden_units=tuple(deep_hidden_units),
                  deep_dropout=tuple(deep_dropout), deep_activation=deep_activation,
                  deep_l2_reg=deep_l2_reg, cross_layer_num=cross_layer_num,
                  cross_layer_l2_reg=cross_layer_l2_reg, use_linear=use_linear,
                  learning_rate=learning_rate, optimizer=optimizer, cross_type=cross_type)
        DeepModel.__init__(self, feat_dict, hp, metrics=eval_metric, epoch=epoch, batch_size=batch_size,
                           random_seed=random_seed,
                           task="classification" if loss_type == "logloss" else "regression",
                           strict_reference=strict_reference, device=device)
        self.what_means_greater, self.use_interactive_session, self.log_dir = (
            what_means_greater, use_interactive_session, log_dir)
        self.loss_type, self.eval_metric = loss_type, eval_metric
        for k, v in hp.items():
            setattr(self, k, v)
