"""DeepFM with the reference's constructor (recman/tf/core/DeepFM.py:30-53) - the class
recman/th/DeepFM.py:12-13 leaves empty."""
from sklearn.metrics import log_loss, roc_auc_score

from .DeepModel import DeepModel


class DeepFM(DeepModel):
    """https://arxiv.org/abs/1703.04247 : final = linear + FM + DNN (DeepFM.py:107-158)."""

    model = "deepfm"

    def __init__(self, feat_dict, embedding_size=8, embedding_l2_reg=0.00001, linear_l2_reg=0.00001,
                 fm_dropout=(1.0, 1.0), deep_hidden_units=(32, 32), deep_dropout=(0.8, 0.8, 0.8),
                 deep_l2_reg=0.00001, deep_activation="relu", epoch=10, batch_size=64,
                 learning_rate=0.001, optimizer="adam", random_seed=2019, use_fm=True, use_deep=True,
                 loss_type="logloss", eval_metric=(roc_auc_score, log_loss), what_means_greater=None,
                 use_interactive_session=False, log_dir="./logs", strict_reference=False,
                 device="cuda"):
        assert use_fm or use_deep  # DeepFM.py:54
        assert loss_type in ["logloss", "mse"], (
            "loss_type can be either 'logloss' for classification task or 'mse' for regression task")
        hp = dict(embedding_size=embedding_size, embedding_l2_reg=embedding_l2_reg,
                  linear_l2_reg=linear_l2_reg, fm_dropout=tuple(fm_dropout),
                  deep_hidden_units=tuple(deep_hidden_units), deep_dropout=tuple(deep_dropout),
                  deep_l2_reg=deep_l2_reg, deep_activation=deep_activation, use_fm=use_fm,
                  use_deep=use_deep, learning_rate=learning_rate, optimizer=optimizer)
        DeepModel.__init__(self, feat_dict, hp, metrics=eval_metric, epoch=epoch, batch_size=batch_size,
                           random_seed=random_seed,
                           task="classification" if loss_type == "logloss" else "regression",
                           strict_reference=strict_reference, device=device)
        # TF-only knobs are accepted and ignored
        self.what_means_greater, self.use_interactive_session, self.log_dir = (
            what_means_greater, use_interactive_session, log_dir)
        self.loss_type, self.eval_metric = loss_type, eval_metric
        for k, v in hp.items():  # sklearn get_params()/clone() need the ctor arguments back
            setattr(self, k, v)
