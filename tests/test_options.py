"""Option surface (SURVEY.md §8b harness row `options.py`): merged YAML trees and the CLI
grammar must equal what the reference's options.py produces (tests/golden/options.json,
captured through the imported reference)."""
import json
import os

import pytest

from conftest import GOLDEN
from matchnerf_amd import options
from matchnerf_amd.edict import to_plain_dict

with open(os.path.join(GOLDEN, "options.json")) as f:
    GOLD = json.load(f)


@pytest.mark.parametrize("name", sorted(GOLD["yaml_trees"]))
def test_yaml_tree_equals_reference(name):
    tree = to_plain_dict(options.load_options(f"configs/{name}.yaml", verbose=False))
    assert json.loads(json.dumps(tree)) == GOLD["yaml_trees"][name]


@pytest.mark.parametrize("case", sorted(GOLD["cli"]))
def test_cli_grammar_equals_reference(case):
    argv = GOLD["cli"][case]["argv"]
    assert json.loads(json.dumps(to_plain_dict(options.parse_arguments(argv)))) == GOLD["cli"][case]["parsed"]


def test_set_merges_and_processes(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cmd = options.parse_arguments(["--yaml=test", "--name=unit", "--nerf.rand_rays_test=4096",
                                   "--nerf.sample_intvs=64", "--n_src_views=3", "--seed=3"])
    opt = options.set(cmd, verbose=False)
    assert opt.nerf.rand_rays_test == 4096 and opt.nerf.sample_intvs == 64
    assert opt.name == "unit_seed3" and opt.output_path == os.path.join("outputs", "unit_seed3")
    assert os.path.isfile(os.path.join(opt.output_path, "run.bash"))
    assert opt.device in ("cpu", "cuda:0")
    options.save_options_file(opt)
    options.save_options_file(opt)  # identical -> no prompt
    assert os.path.isfile(os.path.join(opt.output_path, "options.yaml"))


def test_unknown_cli_key_is_added_off_tty(tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    opt = options.set(options.parse_arguments(["--yaml=test", "--name=u2", "--nerf.brand_new_key=7"]), verbose=False)
    assert opt.nerf.brand_new_key == 7
    assert "not found in original opt" in capsys.readouterr().out


def test_debug_name_truncates_datasets(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    opt = options.set(options.parse_arguments(["--yaml=train", "--name=x_debug"]), verbose=False)
    assert opt.max_epoch == 2 and opt.data_train.max_len == 20 and opt.data_test.dtu.max_len == 1
