"""MammalNet, single-label — the reference's TFAM/train_and_eval_frame_diff_MN.py: CrossEntropy + Accuracy instead of
BCE + micro-mAP (:49,59), everything else as train_and_eval_frame_diff_AK.py."""
from .train_and_eval import Config, ModelTester, ModelTrainer, batches, build_datasets, build_model, main, run, set_seed  # noqa: F401

if __name__ == "__main__":
    main(default_task="singlelabel", default_motion_key="frame_diff")
