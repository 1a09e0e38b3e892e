"""Animal Kingdom, frame-difference motion stream — the reference's TFAM/train_and_eval_frame_diff_AK.py: the same loop as
train_and_eval.py with the ``frame_diff`` batch keys and the YAML configuration (``--config cfg_AK/config_N.yaml``)."""
from .train_and_eval import Config, ModelTester, ModelTrainer, batches, build_datasets, build_model, main, run, set_seed  # noqa: F401

if __name__ == "__main__":
    main(default_task="multilabel", default_motion_key="frame_diff")
