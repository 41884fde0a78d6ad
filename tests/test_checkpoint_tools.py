"""tools/convert_checkpoints.py: whole-module pickles (mapping/train_transformer.py:166, mapping/train_mapping.py:160) ->
plain state_dict files in the same layout, readable with weights_only=True by nested_diffusion_amd.mapping.load_pickled."""
import importlib.util
import os
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("convert_checkpoints", os.path.join(ROOT, "tools", "convert_checkpoints.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


MODULE_SRC = textwrap.dedent('''
    import torch.nn as nn
    class Classifier(nn.Module):                       # stands in for the reference's mlp.Classifier (a class that is NOT
        def __init__(self, n_in=12, num_classes=2):    # importable when the converted file is read back)
            super().__init__()
            self.linear1 = nn.Linear(n_in, 8); self.linear2 = nn.Linear(8, 6)
            self.linear3 = nn.Linear(6, 4); self.linear4 = nn.Linear(4, num_classes)
    class FakeViT(nn.Module):
        def __init__(self):
            super().__init__()
            self.cls_token = nn.Parameter(__import__("torch").zeros(1, 1, 8))
            self.patch_embed = nn.Conv2d(3, 8, 4, 4)
''')


def test_module_pickles_become_plain_state_dicts(tmp_path):
    # 1. write module pickles in the reference's layout from a SEPARATE process, with the defining module on its sys.path only
    src, dst, moddir = tmp_path / "ckpt", tmp_path / "converted", tmp_path / "models"
    (src / "MLPs").mkdir(parents=True); moddir.mkdir()
    (moddir / "fake_mlp.py").write_text(MODULE_SRC)
    writer = textwrap.dedent(f'''
        import sys, torch
        sys.path.insert(0, {str(moddir)!r})
        import fake_mlp
        torch.manual_seed(3)
        torch.save(fake_mlp.FakeViT(), {str(src / "vit_base_patch16_224_ChestXRay.pth")!r})
        for i in range(2):
            torch.save(fake_mlp.Classifier(), {str(src / "MLPs")!r} + f"/block_{{i}}.pth")
        torch.save({{"state_dict": fake_mlp.Classifier().state_dict()}}, {str(src / "MLPs" / "block_2.pth")!r})
    ''')
    subprocess.run([sys.executable, "-c", writer], check=True)
    # the pickles cannot be read here without the class ...
    try:
        torch.load(src / "MLPs" / "block_0.pth", map_location="cpu", weights_only=False)
        unreadable = False
    except Exception:
        unreadable = True
    assert unreadable, "fake_mlp must not be importable in the test process"
    # 2. convert (the tool appends --sys-path for the unpickler, as the reference does at :255)
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_checkpoints.py"), "--src", str(src), "--dst", str(dst),
                         "--sys-path", str(moddir)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr
    assert "MLPs/block_0.pth: 8 tensors" in rc.stdout
    # 3. the converted tree has the same layout and loads with the restricted unpickler, no class needed
    assert sorted(os.listdir(dst / "MLPs")) == ["block_0.pth", "block_1.pth", "block_2.pth"]
    from nested_diffusion_amd.mapping import load_pickled
    sd = load_pickled(str(dst / "MLPs" / "block_1.pth"))
    assert sorted(sd) == sorted(f"linear{i}.{s}" for i in range(1, 5) for s in ("weight", "bias"))
    assert sd["linear1.weight"].shape == (8, 12)
    plain = torch.load(dst / "vit_base_patch16_224_ChestXRay.pth", map_location="cpu", weights_only=True)
    assert set(plain) == {"cls_token", "patch_embed.weight", "patch_embed.bias"}
    assert sorted(load_pickled(str(dst / "MLPs" / "block_2.pth"))) == sorted(sd)          # {'state_dict': ...} form unwrapped


def test_converter_rejects_non_checkpoints(tmp_path):
    tool = _tool()
    bad = tmp_path / "x.pth"
    torch.save([1, 2, 3], bad)
    try:
        tool.convert_file(str(bad), str(tmp_path / "y.pth"))
        raised = False
    except TypeError:
        raised = True
    assert raised
