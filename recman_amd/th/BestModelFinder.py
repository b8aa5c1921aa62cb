"""epoch_callback that keeps (and optionally checkpoints) the best model of a fit().

Mirrors recman/tf/BestModelFinder.py:9-68: called as
`epoch_callback(model=..., eval_results=(train_res, valid_res), df_all=X_train[:1])`
(DeepModel.py:216-221); the score is the FIRST metric of the LAST non-empty result set
(validation when present, else training) and LOWER is better (:46-49).  With
save_model=True the reference writes a tf.train.Checkpoint plus dill pickles of hparams /
feat_dict / df_all into the working directory (:56-68); here the checkpoint is the model's
state_dict under the reference's variable names (DeepModel.save) and the side files are
plain pickles of this package's own objects.
"""
import logging
import os
import pickle

log = logging.getLogger(__name__)


class BestModelFinder:
    def __init__(self, save_model=False, directory="."):
        self._best_score = None
        self._best_eval_results = None
        self._model = None  # (the reference leaves this unset until the first call)
        self.save_model = save_model
        self.directory = directory

    @property
    def best_score(self):
        return self._best_score

    @property
    def best_eval_results(self):
        return self._best_eval_results

    @property
    def best_model(self):
        return self._model

    def __call__(self, **kwargs):
        assert (kwargs["model"] is not None and kwargs["model"].hparams is not None
                and kwargs["model"].feat_dict is not None and kwargs["model"].variables is not None
                and kwargs["eval_results"] is not None and kwargs["df_all"] is not None)  # :29-36
        model = kwargs["model"]
        eval_results = [r for r in kwargs["eval_results"] if r]  # drop the empty validation slot
        score = eval_results[-1][0]
        if self._best_score is None or score < self._best_score:
            log.info("A better model is found!")
            log.info(eval_results)
            self._best_score = score
            self._best_eval_results = eval_results
            self._model = model
            if self.save_model:
                d = self.directory
                model.save(os.path.join(d, "ckpt_model.pt"))
                for name, obj in (("hparams", model.hparams), ("feat_dict", model.feat_dict),
                                  ("df_all", kwargs["df_all"])):
                    with open(os.path.join(d, name), "wb") as w:
                        pickle.dump(obj, w, protocol=pickle.HIGHEST_PROTOCOL)
        return self

    @staticmethod
    def load(model_cls, directory=".", **ctor):
        """Rebuilds the checkpointed model: model_cls(feat_dict, <hparams>, **ctor) + restore."""
        with open(os.path.join(directory, "feat_dict"), "rb") as r:
            feat_dict = pickle.load(r)
        with open(os.path.join(directory, "hparams"), "rb") as r:
            hparams = pickle.load(r)
        model = model_cls.from_hparams(feat_dict, hparams, **ctor)
        model.restore(os.path.join(directory, "ckpt_model.pt"))
        return model
