"""xDeepFM with the reference's constructor (recman/tf/core/xDeepFM.py:26-35): a
FeatureDictionary and a hyper-parameter dict with the keys of hparams.xDeepFM."""
from .DeepModel import DeepModel
from .hparams import xDeepFM as HyperParams


class xDeepFM(DeepModel):
    """https://arxiv.org/pdf/1803.05170.pdf : final = linear + CIN + DNN (xDeepFM.py:47-104)."""

    model = "xdeepfm"

    def __init__(self, feat_dict, hparams: dict, task="classification", metrics=(), epoch=10,
                 batch_size=64, random_seed=2019, strict_reference=False, device="cuda"):
        hp = HyperParams().defaults()
        hp.update(hparams)
        DeepModel.__init__(self, feat_dict=feat_dict, hparams=hp, epoch=epoch, batch_size=batch_size,
                           random_seed=random_seed, metrics=metrics, task=task,
                           strict_reference=strict_reference, device=device)
