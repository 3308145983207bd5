"""Flags of the RLCF classification entry points: same names, types and defaults as the reference (TPT/params.py:13-98), so the command
lines of TPT/scripts/*.sh parse identically.  Flags the HIP path has no use for (--workers, --dataset_mode, --cocoop, --corruption,
--level, --kd_loss, --confidence_gap*) are accepted and ignored; --hard_aug 1 selects the BYOL-style recipe of the view pipeline (datautils.HardAugParams)."""
import argparse


def none_or_str(value):
    """TPT/params.py:8-11"""
    return None if value == "None" else value


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RLCF test-time adaptation (MI355X-native hot path)")
    p.add_argument("data", nargs="?", default="synthetic", help="dataset root (the HIP path ships a synthetic stream only)")
    p.add_argument("--test_sets", type=str, default="A/R/V/K/I")
    p.add_argument("-a", "--arch", default="RN50")
    p.add_argument("--resolution", default=224, type=int)
    p.add_argument("-b", "--batch-size", "--batch_size", dest="batch_size", default=64, type=int)     # (the scripts write -b)
    p.add_argument("--lr", "--learning-rate", default=5e-3, type=float, dest="lr")
    p.add_argument("--weight_decay", default=5e-4, type=float)
    p.add_argument("-p", "--print-freq", "--print_freq", dest="print_freq", default=500, type=int)
    p.add_argument("--gpu", default=0, type=int)
    p.add_argument("--tpt", action="store_true", default=False)
    p.add_argument("--selection_p", default=0.1, type=float)
    p.add_argument("--tta_steps", default=1, type=int)
    p.add_argument("--n_ctx", default=4, type=int)
    p.add_argument("--ctx_init", default=None, type=str)
    p.add_argument("--cocoop", action="store_true", default=False, help="accepted for script compatibility (CoCoOp initialisation is not on the RLCF path)")
    p.add_argument("--load", default=None, type=none_or_str)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--output", type=str, default="exp_01")
    p.add_argument("--dataset_mode", type=str, default="test", help="accepted (the HIP path ships a synthetic stream only)")
    p.add_argument("--workers", default=8, type=int, help="accepted (views are generated on the device: no loader workers)")
    p.add_argument("--hard_aug", type=int, default=0, help="BYOL-style recipe of the view pipeline (datautils.py:77-87)")
    p.add_argument("--confidence_gap", type=int, default=0, help="experimental in the reference, never read by its loop: accepted, ignored")
    p.add_argument("--confidence_gap_w", type=float, default=0.5)
    p.add_argument("--corruption", type=str, default="defocus_blur")
    p.add_argument("--level", type=str, default="5")
    p.add_argument("--kd_loss", type=str, default="KD", choices=["KD", "DKD", "ATKD"])
    p.add_argument("--sample_k", type=int, default=5)
    p.add_argument("--multiple_reward_models", type=int, default=0)
    p.add_argument("--reward_arch", type=str, default="ViT-L/14")
    p.add_argument("--reward_process", type=int, default=1)
    p.add_argument("--process_batch", type=int, default=0)
    p.add_argument("--reward_amplify", type=int, default=0)
    p.add_argument("--weighted_scores", type=int, default=1)
    p.add_argument("--min_entropy_reg", type=int, default=0)
    p.add_argument("--min_entropy_w", type=float, default=0.1)
    p.add_argument("--momentum_update", type=int, default=0, help="update the image encoder in a momentum fashion over test samples")
    p.add_argument("--update_freq", type=int, default=256)
    p.add_argument("--update_w", type=float, default=1.0)
    p.add_argument("--tta_momentum", type=float, default=0.9999)
    p.add_argument("--tune_norm", type=int, default=0)
    p.add_argument("--augmix", type=int, default=1,
                   help="AugMix op chains in the view pipeline; as in the reference (tpt_cls_rl.py:150) only for the fine-grained sets (len(set_id) > 1)")
    p.add_argument("--prior_strength", type=int, default=-1, help="ResNet student: >= 0 blends running and batch BatchNorm statistics with prior s/(s+1) (tune_cls_rl.py:35-44); -1: train-mode BatchNorm")
    # not in the reference's parser (default = unset -> the environment, then 1 = the reference's loop): tpt_cls_rl._loop_option
    p.add_argument("--images_per_pass", type=int, default=None, help="test images handed to the engine per call (RLCF_IMAGES_PER_PASS; 1 = the reference's loop)")
    p.add_argument("--in_flight", type=int, default=None, help="one image per engine call, this many samples side by side on their own engines / streams (RLCF_IN_FLIGHT)")
    p.add_argument("--clip_root", type=str, default="", help="directory of OpenAI-layout state dicts (<arch>.pt)")
    return p.parse_args(argv)
