"""Dev tool (build container only): write the remaining phase-1 / phase-3 YAML configs from the
VALUES of the reference's files (hyper-parameters are data; layout and comments are ours).
    python tools/make_configs.py /root/reference
"""
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P3_ORDER = ["batch_size", "num_train", "num_epochs", "seq_length", "window_size", "nb_samples", "gamma", "beta", "eta",
            "lr_gen", "lr_critic", "n_critic_steps", "freeze_epoch", "enc_type", "activ", "input_vector_size",
            "latent_vector_size", "noise_size", "n_cells", "nblocks_gen", "size", "output_size", "ablated", "channels",
            "code_size", "init_kernel", "nblocks_critic", "dance_types", "dataset", "folder"]
P1_ORDER = ["batch_size", "num_train", "num_epochs", "gamma", "lr_gen", "lr_critic", "n_critic_steps",
            "latent_vector_size", "size", "output_size", "nblocks_gen", "nblocks_critic", "folder"]
P3_NOTE = {"l1_enhanced": "stronger L1 reconstruction weight", "l1_only": "L1 weight as in default (ablation label of the reference)",
           "minimal": "0.04 s audio window (640 samples): too short for the default encoder (SURVEY.md A.1)",
           "noise_enhanced": "larger generator noise", "tv": "total-variation regulariser on"}


def dump(d, order, header, path):
    keys = [k for k in order if k in d] + [k for k in d if k not in order]
    with open(path, "w") as f:
        f.write(header)
        for k in keys:
            v = d[k]
            if isinstance(v, dict):
                f.write("%s:\n" % k)
                for kk, vv in v.items():
                    f.write("  " + yaml.safe_dump({kk: vv}, default_flow_style=False))
            else:
                f.write(yaml.safe_dump({k: v}, default_flow_style=isinstance(v, list) or None).replace("{", "").replace("}", "")
                        if isinstance(v, list) else yaml.safe_dump({k: v}, default_flow_style=False))


def main(ref):
    for name in sorted(os.listdir(os.path.join(ref, "phase3/configs"))):
        out = os.path.join(ROOT, "music2dance_amd/phase3/configs", name)
        if os.path.exists(out):
            continue
        d = yaml.safe_load(open(os.path.join(ref, "phase3/configs", name)))
        stem = name[:-5]
        dump(d, P3_ORDER, "# Phase 3 - %s: %s.\n# Same keys / values as the reference's phase3/configs/%s.\n"
             % (stem, P3_NOTE.get(stem, "variant"), name), out)
    for name in sorted(os.listdir(os.path.join(ref, "phase1/configs"))):
        out = os.path.join(ROOT, "music2dance_amd/phase1/configs", name)
        if os.path.exists(out):
            continue
        d = yaml.safe_load(open(os.path.join(ref, "phase1/configs", name)))
        dump(d, P1_ORDER, "# Phase 1 - still-pose residual-MLP WGAN-GP: %d block(s), latent %d, width %d\n"
             "# (same keys / values as the reference's phase1/configs/%s).\n"
             % (d["nblocks_gen"], d["latent_vector_size"], d["size"], name), out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
