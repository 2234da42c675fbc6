"""Option system of the hot path — same surface as the reference's options.py.

Mirrors /root/reference/options.py:19-160: ``parse_arguments`` (``--a.b.c=value`` grammar),
``set`` (YAML + command line), ``load_options`` (``_parent_`` inheritance),
``override_options``, ``process_options`` (seeds, device, output dir) and
``save_options_file``.  Differences, all deliberate:

* the reference asks on stdin whether to add an unknown command-line key or to overwrite an
  existing options file (options.py:86-93, 149-154); off a tty that blocks forever, so here
  the question is only asked on an interactive terminal — otherwise the key is added / the
  file overwritten and a note is printed;
* ``configs/<name>.yaml`` is looked up in the working directory first (as the reference does)
  and then in this package's own ``configs/`` directory, so the tool works from anywhere;
* ``opt.device``: the render path is HIP-only.  ``cpu: true`` or a box without a GPU yields
  ``"cpu"`` exactly as in the reference (options.py:133) and the model then refuses to render
  instead of silently falling back.
"""
import os
import random
import string
import sys
import time

import numpy as np
import yaml

from .edict import EasyDict as edict
from .edict import to_plain_dict

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))


def parse_arguments(args):
    """``--k1.k2=v`` -> nested dict.  ``--k`` -> True, ``--k!`` -> False, ``--k=`` -> None,
    values go through ``yaml.safe_load``; a string value containing ',' becomes a list
    (digits -> int, empty items dropped).  (options.py:19-47)"""
    tree = {}
    for arg in args:
        if not arg.startswith("--"):
            raise AssertionError(f"options must start with '--': {arg}")
        body = arg[2:]
        if "=" in body:
            key_path, raw = body.split("=")
        elif body.endswith("!"):
            key_path, raw = body[:-1], "false"
        else:
            key_path, raw = body, "true"
        *parents, leaf = key_path.split(".")
        node = tree
        for k in parents:
            node = node.setdefault(k, {})
        if leaf in node:
            raise AssertionError(leaf)
        value = yaml.safe_load(raw)
        if isinstance(value, str) and "," in value:
            value = [int(x) if x.isdigit() else x for x in value.split(",") if x.strip()]
        node[leaf] = value
    return edict(tree)


def _resolve(fname):
    if os.path.isfile(fname):
        return fname
    alt = os.path.join(_PKG_DIR, fname)
    if os.path.isfile(alt):
        return alt
    alt = os.path.join(_PKG_DIR, "configs", os.path.basename(fname))
    if os.path.isfile(alt):
        return alt
    raise FileNotFoundError(fname)


def load_options(fname, verbose=True):
    """YAML -> option tree, parents first (options.py:63-76)."""
    with open(_resolve(fname)) as f:
        opt = edict(yaml.safe_load(f))
    if "_parent_" in opt:
        parents = opt.pop("_parent_")
        for parent in ([parents] if isinstance(parents, str) else parents):
            opt = override_options(load_options(parent, verbose), opt, key_stack=[])
    if verbose:
        print("loading {}...".format(fname))
    return opt


def _ask(question):
    if not sys.stdin.isatty():
        print(f"{question} -> yes (non-interactive run)")
        return True
    answer = None
    while answer not in ("y", "n"):
        answer = input(question + " (y/n) ")
    return answer == "y"


def override_options(opt, opt_over, key_stack=None, safe_check=False):
    """Recursive merge of ``opt_over`` into ``opt`` (options.py:79-95)."""
    key_stack = key_stack or []
    for key, value in opt_over.items():
        if isinstance(value, dict):
            child = opt.get(key, None)
            if not isinstance(child, dict):
                child = edict()
            opt[key] = override_options(child, value, key_stack + [key], safe_check)
        else:
            if safe_check and key not in opt:
                name = ".".join(key_stack + [key])
                if not _ask(f"\"{name}\" not found in original opt, add?"):
                    print("safe exiting...")
                    sys.exit()
            opt[key] = value
    return opt


def process_options(opt, make_output_dir=True):
    """Run name, debug truncation, seeding, output path, device (options.py:98-133)."""
    import torch
    if opt.name is None:
        opt.name = time.strftime("%b%d_%H%M%S").lower()
    if isinstance(getattr(opt, "gpu_ids"), int):
        opt.gpu_ids = [opt.gpu_ids]
    if "_debug" in str(opt.name):
        if hasattr(opt, "data_train"):
            opt.data_train.max_len = 20
        if hasattr(opt, "data_val"):
            opt.data_val.max_len = 1
        if hasattr(opt, "data_test"):
            for x in opt.data_test:
                if opt.data_test[x] is not None:
                    opt.data_test[x].max_len = 1
        opt.max_epoch = 2
    if opt.seed is not None:
        random.seed(opt.seed)
        np.random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(opt.seed)
        if opt.seed != 0:
            opt.name = str(opt.name) + "_seed{}".format(opt.seed)
    else:
        opt.name = str(opt.name) + "_" + "".join(random.choice(string.ascii_uppercase) for _ in range(4))
    opt.output_path = os.path.join(opt.output_root, opt.name)
    if make_output_dir:
        os.makedirs(opt.output_path, exist_ok=True)
        with open(os.path.join(opt.output_path, "run.bash"), "a+") as f:
            f.write("python %s\n" % (" ".join(sys.argv)))
    local_rank = int(os.environ.get("LOCAL_RANK", -1))
    ordinal = local_rank if local_rank >= 0 else opt.gpu_ids[0]
    opt.device = "cpu" if opt.cpu or not torch.cuda.is_available() else "cuda:{}".format(ordinal)
    return opt


def set(opt_cmd=None, make_output_dir=True, verbose=True):  # noqa: A001 - reference name
    """YAML named by ``--yaml`` merged with the command line, then processed (options.py:50-60)."""
    opt_cmd = opt_cmd or {}
    assert "yaml" in opt_cmd, "--yaml=<config name> is required"
    opt = load_options("configs/{}.yaml".format(opt_cmd["yaml"]), verbose)
    opt = override_options(opt, opt_cmd, key_stack=[], safe_check=True)
    process_options(opt, make_output_dir)
    return opt


def save_options_file(opt):
    """Write ``<output_path>/options.yaml`` (options.py:136-160)."""
    path = "{}/options.yaml".format(opt.output_path)
    plain = to_plain_dict(opt)
    if os.path.isfile(path):
        with open(path) as f:
            old = yaml.safe_load(f)
        if old == plain:
            print("existing options file found (identical)")
        elif not _ask("existing options file found (different from current one), override?"):
            print("safe exiting...")
            sys.exit()
    else:
        print("(creating new options file...)")
    with open(path, "w") as f:
        yaml.safe_dump(plain, f, default_flow_style=False, indent=4)
