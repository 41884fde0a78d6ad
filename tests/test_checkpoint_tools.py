"""The reference's whole-module checkpoint pickles (mapping/train_transformer.py:166, mapping/train_mapping.py:160) read WITHOUT the
classes that defined them: directly by nested_diffusion_amd.mapping.load_pickled (skeleton unpickler), and through
tools/convert_checkpoints.py (-> plain state_dict files in the same layout, readable with weights_only=True)."""
import importlib.util
import os
import subprocess
import sys
import textwrap

import pytest
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


# a module tree shaped like the reference's pickles: nested containers, buffers (one non-persistent), plain python attributes --
# defined in a module that only the WRITING process can import, under a package path like timm's
TREE_SRC = textwrap.dedent('''
    import torch, torch.nn as nn
    class Attention(nn.Module):
        def __init__(self, dim=8, heads=2):
            super().__init__()
            self.num_heads, self.scale = heads, (dim // heads) ** -0.5
            self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
            self.attn_drop = nn.Dropout(0.0)
    class Block(nn.Module):
        def __init__(self, dim=8):
            super().__init__()
            self.norm1, self.attn = nn.LayerNorm(dim, eps=1e-6), Attention(dim)
            self.drop_path = nn.Identity()
            self.mlp = nn.Sequential(nn.Linear(dim, 16), nn.GELU(), nn.Linear(16, dim))
    class VisionTransformer(nn.Module):
        def __init__(self, dim=8):
            super().__init__()
            self.num_features = self.embed_dim = dim
            self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
            self.patch_embed = nn.Conv2d(3, dim, 4, 4)
            self.blocks = nn.Sequential(Block(dim), Block(dim))
            self.bn = nn.BatchNorm1d(dim)                          # buffers: running_mean / running_var / num_batches_tracked
            self.register_buffer("scratch", torch.zeros(3), persistent=False)
            self.head = nn.Linear(dim, 2)
''')


def _write_tree(tmp_path):
    pkg = tmp_path / "site" / "timmlike" / "models"
    pkg.mkdir(parents=True)
    (tmp_path / "site" / "timmlike" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "vision_transformer.py").write_text(TREE_SRC)
    writer = textwrap.dedent(f'''
        import sys, torch
        sys.path.insert(0, {str(tmp_path / "site")!r})
        from timmlike.models.vision_transformer import VisionTransformer
        torch.manual_seed(11)
        m = VisionTransformer().eval()
        torch.save(m, {str(tmp_path / "module.pth")!r})                                          # zip format (torch >= 1.6 default)
        torch.save(m, {str(tmp_path / "module_legacy.pth")!r}, _use_new_zipfile_serialization=False)
        torch.save(m.state_dict(), {str(tmp_path / "state.pth")!r})
    ''')
    subprocess.run([sys.executable, "-c", writer], check=True)


def test_module_pickles_load_directly_without_their_classes(tmp_path):
    """mapping.load_pickled on a whole-module pickle whose classes exist nowhere in this process: same names, order and values as the
    module's own state_dict() (nested sub-modules, parameters, persistent buffers; the non-persistent buffer left out)."""
    import warnings
    from nested_diffusion_amd.mapping import load_pickled
    _write_tree(tmp_path)
    assert "timmlike" not in sys.modules
    want = torch.load(tmp_path / "state.pth", map_location="cpu", weights_only=True)
    assert "blocks.1.attn.qkv.weight" in want and "bn.running_var" in want and "scratch" not in want
    for name in ("module.pth", "module_legacy.pth"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                        # legacy format: "couldn't retrieve source code for container"
            got = load_pickled(str(tmp_path / name))
        assert list(got) == list(want), name
        for k in want:
            assert torch.equal(got[k], want[k]) and got[k].dtype == want[k].dtype, (name, k)
    assert "timmlike" not in sys.modules                           # nothing was imported to do it


def test_skeleton_unpickler_still_refuses_other_globals(tmp_path):
    """Only module trees get the lenient treatment: a pickle that names a standard-library / builtins / torch global outside
    torch's weights_only allow-list is refused before anything is called."""
    import pickle
    from nested_diffusion_amd.mapping import load_pickled
    marker = tmp_path / "executed"

    class RunsACommand:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class Evals:
        def __reduce__(self):
            return (eval, ("1+1",))

    class CallsTorchHub:
        def __reduce__(self):
            return (torch.hub.load, ("x", "y"))

    for i, bad in enumerate((RunsACommand(), Evals(), CallsTorchHub())):
        path = tmp_path / f"bad{i}.pth"
        torch.save({"weights": torch.zeros(2), "extra": bad}, path)
        with pytest.raises(pickle.UnpicklingError, match="refusing global"):
            load_pickled(str(path))
    assert not marker.exists()


def test_skeleton_unpickler_reads_pickles_shaped_like_old_torch_writes_them(tmp_path):
    """A module pickled by torch 1.10 (requirements.txt:59) lacks attribute tables newer torch versions add in __init__
    (`_non_persistent_buffers_set` exists since 1.6, the `*_with_kwargs` / `_state_dict_pre_hooks` tables do not), and timm-style
    modules may keep a functools.partial of a torch class as an attribute: neither may stop the tensors from being read."""
    from nested_diffusion_amd.mapping import load_pickled
    site = tmp_path / "site"
    site.mkdir()
    (site / "oldstyle.py").write_text(textwrap.dedent('''
        import functools, torch, torch.nn as nn
        class Leaf(nn.Module):
            def __init__(self):
                super().__init__()
                self.fc = nn.Linear(4, 3)
                self.norm_layer = functools.partial(nn.LayerNorm, eps=1e-6)     # a callable kept as an attribute
                self.register_buffer("running", torch.arange(3.0))
        class Root(nn.Module):
            def __init__(self):
                super().__init__()
                self.blocks = nn.ModuleList([Leaf(), Leaf()])
                self.head = nn.Linear(3, 2)
    '''))
    writer = textwrap.dedent(f'''
        import sys, torch
        sys.path.insert(0, {str(site)!r})
        import oldstyle
        torch.manual_seed(5)
        m = oldstyle.Root().eval()
        torch.save(m.state_dict(), {str(tmp_path / "state.pth")!r})
        for mod in m.modules():                      # strip what a torch 1.10 pickle would not contain
            for k in list(mod.__dict__):
                if k.endswith("_with_kwargs") or k.endswith("_always_called") or k in ("_state_dict_pre_hooks", "_load_state_dict_post_hooks",
                                                                                     "_forward_pre_hooks_with_kwargs", "_backward_pre_hooks", "_is_full_backward_hook"):
                    del mod.__dict__[k]
        torch.save(m, {str(tmp_path / "old.pth")!r})
    ''')
    subprocess.run([sys.executable, "-c", writer], check=True)
    want = torch.load(tmp_path / "state.pth", map_location="cpu", weights_only=True)
    got = load_pickled(str(tmp_path / "old.pth"))
    assert list(got) == list(want) and all(torch.equal(got[k], want[k]) for k in want)
    assert "oldstyle" not in sys.modules


def test_skeleton_unpickler_never_calls_a_real_torch_nn_constructor(tmp_path):
    """The stock pickle machine leaves REDUCE unrestricted for any global find_class resolves.  A file that asks for
    torch.nn.Linear(10**9, 10**9) (4 EB of weights: an allocation attack, no code execution) must not reach the real class: every
    torch.nn.modules class resolves to an inert stub whose constructor takes no arguments."""
    import pickle
    from nested_diffusion_amd.mapping import _SkeletonUnpickler, load_pickled

    class AllocatesALot:
        def __reduce__(self):
            return (torch.nn.Linear, (10 ** 9, 10 ** 9))

    path = tmp_path / "alloc.pth"
    torch.save({"weights": torch.zeros(2), "extra": AllocatesALot()}, path)
    with pytest.raises((TypeError, pickle.UnpicklingError)):
        load_pickled(str(path))
    import io
    stub = _SkeletonUnpickler(io.BytesIO(b"")).find_class("torch.nn.modules.linear", "Linear")
    assert stub is not torch.nn.Linear and issubclass(stub, torch.nn.Module) and stub.__name__ == "Linear"
    assert _SkeletonUnpickler._allowed() is _SkeletonUnpickler._allowed()          # the allow-list is built once
