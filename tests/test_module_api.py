"""CPU: the drop-in module surface (names, shapes, assertions, init stream) without touching a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from vit_amd import _lib
from vit_amd import surgical_world_model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vit_amd.h")).read()
    declared = set(re.findall(r"\b(va_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), "python binding table and header disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vit_amd.h but not exported"
    assert _lib.load().va_version() >= 100


def test_encoder_state_dict_matches_reference_layout():
    enc = M.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=2, num_heads=16)
    sd = enc.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if "inv_freq" not in k}
    assert shapes == R.encoder_param_shapes(2)
    assert list(shapes) == enc._param_order
    assert tuple(sd["rope.inv_freq_t"].shape) == (128,) and tuple(sd["rope.inv_freq_h"].shape) == (192,)
    assert torch.equal(sd["rope.inv_freq_w"], R.rope_inv_freq(384))


def test_seed_reproduces_reference_initialisation(golden_dir):
    g = np.load(os.path.join(golden_dir, "seed_canary.npz"), allow_pickle=False)
    torch.manual_seed(0)
    enc = M.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=1, num_heads=16)
    assert list(enc.state_dict().keys()) == [str(k) for k in g["keys"]]
    np.testing.assert_allclose(enc.tubelet_embed.projection.weight.double().sum().item(), g["conv_w_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(enc.blocks[0].attn.qkv_proj.weight.double().sum().item(), g["qkv_w_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(enc.blocks[0].mlp[0].weight.double().sum().item(), g["fc1_w_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(enc.blocks[0].mlp[3].bias.double().sum().item(), g["fc2_b_sum"], rtol=0, atol=1e-9)


def test_constructor_assertions_match_reference():
    with pytest.raises(AssertionError):       # 256+384+384 != 384
        M.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=384, depth=1, num_heads=6)
    with pytest.raises(AssertionError):       # 1024 % 12 != 0  (the reference's own default crashes the same way)
        M.VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=1)
    with pytest.raises(ValueError):
        M.TubeletEmbedding().inflate_weights_from_2d(torch.zeros(3, 3, 3))
    te = M.TubeletEmbedding(img_size=32, num_frames=4)
    te.inflate_weights_from_2d(torch.ones(1024, 3, 16, 16))
    assert te.projection.weight[:, :, 0].abs().sum() == 0 and (te.projection.weight[:, :, 1] == 1).all()


def test_world_model_surface():
    m = M.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=1, encoder_heads=16,
                                              predictor_depth=3, predictor_heads=8)
    assert m.encoder_dim == 1024 and m.predictor_dim == 1024 and m.tokens_per_frame == 4
    assert m.encoder.tubelet_embed.num_temporal_patches == 2
    keys = set(m.state_dict())
    for k in ("video_proj.weight", "action_state_embed.inv_freq_t", "action_state_embed.action_proj.weight",
              "predictor_blocks.0.spatial_attn.in_proj_weight", "predictor_blocks.0.spatial_attn.out_proj.bias",
              "predictor_blocks.0.temporal_layer.q_conv1d.weight", "predictor_blocks.1.temporal_layer.o_norm.bias",
              "predictor_blocks.2.mlp.2.bias", "predictor_norm.weight", "predictor_to_encoder.weight",
              "sterility_classifier.current_classifier.4.weight", "sterility_classifier.future_classifier.1.bias",
              "autoregressive_predictor.spatial_pooling.queries",
              "autoregressive_predictor.spatial_pooling.attention.in_proj_weight",
              "autoregressive_predictor.spatial_predictor.4.weight", "autoregressive_predictor.to_encoder_features.2.bias",
              "autoregressive_predictor.action_predictor.5.weight", "autoregressive_predictor.state_uncertainty.2.bias"):
        assert k in keys, k
    # use_deltanet = i < depth*2//3  -> blocks 0,1 of 3
    assert "predictor_blocks.2.temporal_layer.q_proj.weight" not in keys
    # encoder on CPU must refuse loudly instead of silently computing somewhere else
    with pytest.raises(_lib.VitAmdError):
        m.encode_video(torch.zeros(1, 3, 4, 32, 32))


def test_block_causal_mask_and_interleave():
    mask = M.BlockCausalMask.create_mask(12, 4, "cpu")
    ref = torch.ones(12, 12, dtype=torch.bool)
    for t in range(3):
        ref[t * 4:(t + 1) * 4, :(t + 1) * 4] = False
    assert torch.equal(mask, ref)
    mask = M.BlockCausalMask.create_mask(10, 4, "cpu")
    ref = torch.ones(10, 10, dtype=torch.bool)
    for t in range(2):
        ref[t * 4:(t + 1) * 4, :(t + 1) * 4] = False
    assert torch.equal(mask, ref)
    m = M.SurgicalActionConditionedWorldModel.__new__(M.SurgicalActionConditionedWorldModel)
    v = torch.arange(2 * 3 * 2 * 5).float().view(2, 3, 2, 5)
    a, s = -torch.ones(2, 3, 5), -2 * torch.ones(2, 3, 5)
    out = M.SurgicalActionConditionedWorldModel.interleave_tokens(m, v, a, s)
    assert out.shape == (2, 12, 5)
    assert torch.equal(out[:, 0:2], v[:, 0]) and (out[:, 2] == -1).all() and (out[:, 3] == -2).all()
    assert torch.equal(out[:, 4:6], v[:, 1])


def test_predictor_pieces_on_cpu():
    """The predictor runs as torch ops, so its arithmetic can be exercised on CPU: the rank-1 delta rule
    must equal the dense matrix form of the recurrence."""
    torch.manual_seed(0)
    layer = M.GatedDeltaLayer(64, num_heads=2, head_dim=32).eval()
    B, L, H, D = 2, 7, 2, 32
    q, k, v = (torch.nn.functional.normalize(torch.randn(B, L, H, D), dim=-1) for _ in range(3))
    alpha, beta = torch.rand(B, L, H), torch.rand(B, L, H)
    out, state = layer._sequential_gated_delta(q, k, v, alpha, beta, None)
    S = torch.zeros(B, H, D, D)
    eye = torch.eye(D)
    for t in range(L):
        kk = k[:, t, :, :, None] @ k[:, t, :, None, :]
        S = S @ (alpha[:, t, :, None, None] * (eye - beta[:, t, :, None, None] * kk)) + \
            beta[:, t, :, None, None] * (v[:, t, :, :, None] @ k[:, t, :, None, :])
        np.testing.assert_allclose(out[:, t].numpy(), (S @ q[:, t, :, :, None]).squeeze(-1).numpy(), atol=1e-5)
    np.testing.assert_allclose(state.numpy(), S.numpy(), atol=1e-5)
    y, st = layer(torch.randn(B, L, 64))
    assert y.shape == (B, L, 64) and st.shape == (B, 2, 32, 32)


_IMPORT_FORMS = (
    "from surgical_world_model import SurgicalActionConditionedWorldModel",                     # s1:26, ft:40
    "from surgical_world_model import SurgicalActionConditionedWorldModel, VideoViTEncoder",    # lp:23
    "import surgical_world_model",
)


def _reference_layout(tmp_path):
    """A directory laid out like the reference's: a script next to a `surgical_world_model.py` of its own (a decoy
    here) and a sibling module that imports the model too, as stage1_vjepa2_finetuning.py imports
    stage1_surgvu_pretraining.py."""
    (tmp_path / "surgical_world_model.py").write_text(
        "class SurgicalActionConditionedWorldModel: pass\nclass VideoViTEncoder: pass\nDECOY = True\n")
    (tmp_path / "sibling_stage1.py").write_text(
        "from surgical_world_model import SurgicalActionConditionedWorldModel as SiblingModel\n")
    script = tmp_path / "train_script.py"
    script.write_text(
        "import json, os, sys\n" + "\n".join(_IMPORT_FORMS) + "\n"
        "from sibling_stage1 import SiblingModel\n"
        "print(json.dumps({'cls': SurgicalActionConditionedWorldModel.__module__, 'enc': VideoViTEncoder.__module__,\n"
        "                  'sibling': SiblingModel.__module__, 'file': surgical_world_model.__file__,\n"
        "                  'decoy': hasattr(surgical_world_model, 'DECOY'), 'argv': sys.argv[1:], 'name': __name__,\n"
        "                  'path0': sys.path[0], 'all': sorted(getattr(surgical_world_model, '__all__', []))}))\n")
    return script


def _child(cmd, env_extra, cwd):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(env_extra)
    out = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_launcher_runs_an_unchanged_script_on_the_drop_in(tmp_path):
    """`python -m vit_amd.run script.py args` — the documented way to run the reference's scripts unchanged — must
    resolve every import form of the scripts (stage1_surgvu_pretraining.py:26, stage1_vjepa2_finetuning.py:40,
    load_vjepa2_pretrained.py:23), also from a sibling module, to this package although a `surgical_world_model.py`
    lies next to the script, and must hand the script its arguments, `__main__` and its own directory on the path."""
    import sys
    script = _reference_layout(tmp_path)
    info = _child([sys.executable, "-m", "vit_amd.run", str(script), "--epochs", "3"], {"PYTHONPATH": ROOT}, cwd="/")
    assert info["cls"] == info["enc"] == info["sibling"] == "vit_amd.surgical_world_model"
    assert info["file"] == os.path.join(ROOT, "vit_amd", "surgical_world_model.py") and not info["decoy"]
    assert info["argv"] == ["--epochs", "3"] and info["name"] == "__main__" and info["path0"] == str(tmp_path)
    for name in ("rotate_half", "apply_rotary_pos_emb", "TubeletEmbedding", "VideoRoPE3D", "RoPEMultiheadAttention",
                 "GatedDeltaLayer", "VideoViTBlock", "VideoViTEncoder", "ActionStateEmbedding", "HybridPredictorBlock",
                 "BlockCausalMask", "SpatialAttentionPooling", "AutoregressivePredictor", "DualSterilityClassifier",
                 "SurgicalActionConditionedWorldModel"):
        assert name in info["all"], name


def test_a_path_entry_alone_does_not_win_over_the_script_directory(tmp_path):
    """Why the launcher exists: with only `dropin/` on PYTHONPATH a script launched as a file still gets the
    `surgical_world_model.py` that lies next to it (the script's directory is searched first).  The documentation must
    never go back to recommending that for scripts inside the reference's directory."""
    import sys
    script = _reference_layout(tmp_path)
    info = _child([sys.executable, str(script)], {"PYTHONPATH": os.path.join(ROOT, "dropin")}, cwd="/")
    assert info["decoy"] and info["cls"] == "surgical_world_model"


def test_launcher_rejects_a_missing_script():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-m", "vit_amd.run", "/nonexistent/script.py"], env=env, cwd="/",
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "no such script" in out.stderr


def test_shim_directory_serves_imports_that_have_no_model_file_beside_them(golden_dir):
    """`dropin/` on the path is enough wherever no `surgical_world_model.py` shadows it: `python -c`, `python -m`, a
    notebook, or a copy of a script kept outside the reference's directory."""
    import sys
    code = (
        "import sys, json\n"
        "from surgical_world_model import SurgicalActionConditionedWorldModel, VideoViTEncoder\n"
        "import surgical_world_model as m\n"
        "import torch\n"
        "torch.manual_seed(0)\n"
        "enc = VideoViTEncoder(img_size=32, num_frames=4, embed_dim=1024, depth=1, num_heads=16)\n"
        "print(json.dumps({'file': m.__file__, 'cls': SurgicalActionConditionedWorldModel.__module__,\n"
        "                  'keys': list(enc.state_dict().keys())}))\n")
    info = _child([sys.executable, "-c", code], {"PYTHONPATH": os.path.join(ROOT, "dropin")}, cwd="/")
    assert info["file"] == os.path.join(ROOT, "dropin", "surgical_world_model.py")
    assert info["cls"] == "vit_amd.surgical_world_model"
    g = np.load(os.path.join(golden_dir, "seed_canary.npz"), allow_pickle=False)
    assert info["keys"] == [str(k) for k in g["keys"]]


def test_rotary_helpers_match_their_definition():
    x = torch.arange(8.0).view(1, 8)
    assert torch.equal(M.rotate_half(x), torch.tensor([[-4.0, -5, -6, -7, 0, 1, 2, 3]]))
    cos, sin = torch.full((1, 8), 0.5), torch.full((1, 8), 2.0)
    assert torch.equal(M.apply_rotary_pos_emb(x, cos, sin), x * cos + M.rotate_half(x) * sin)
    att = M.RoPEMultiheadAttention(1024, 16, rope_dims={"t_dim": 256, "h_dim": 384, "w_dim": 384})
    grid = (2, 2, 2)
    cache = M.VideoRoPE3D(1024)(*grid)
    enc = M.VideoViTEncoder.__new__(M.VideoViTEncoder)
    pos = M.VideoViTEncoder._create_token_positions(enc, *grid, device="cpu")
    q = R.hashed_uniform("api.rope.q", (2, 8, 1024), 1.0)
    torch.testing.assert_close(att.apply_rope_to_qk(q, cache, pos), R.rope_rotate(q, *grid), rtol=1e-6, atol=1e-6)
    # stand-alone forwards refuse CPU tensors instead of computing somewhere else
    for call in (lambda: att(q), lambda: M.TubeletEmbedding(32, 4)(torch.zeros(1, 3, 4, 32, 32)),
                 lambda: M.VideoViTBlock(1024, 16)(q)):
        with pytest.raises(_lib.VitAmdError):
            call()


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """include/vit_amd.h is the C-ABI boundary: a C99 translation unit (-pedantic) that includes it must compile, link
    against vit_amd/libvit_amd.so and call the entry points that need no device."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    from vit_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "vit_amd.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '    printf("%d %lld %lld\\n", va_version(), (long long)va_attn_mask_words(200),\n'
                   '           (long long)va_attn_bwd_workspace_bytes(2, 64, 16));\n'
                   '    return va_last_error() != 0 ? 0 : 1;\n}\n')
    exe = tmp_path / "abi"
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                    "-o", str(exe), "-L", lib_dir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) >= 100
    assert int(out[1]) == 200 * 2 * 4          # N * 2 * ceil(N / 64) words per [N, N] mask
    assert int(out[2]) == 5 * 2 * 16 * 64 * 4  # five floats per (batch, head, query)


def test_every_entry_point_rejects_null_arguments_with_a_message():
    """Nothing may cross the C boundary as a crash or an exception: every compute entry point called with null
    pointers and zero sizes returns a negative code and leaves a message naming itself (run in a child process, so a
    crash would fail this test instead of ending the test run).  No device is needed: the checks come first."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "from vit_amd import _lib\n"
        "lib = _lib.load()\n"
        "for name, (res, args) in _lib.SIGNATURES.items():\n"
        "    if name in ('va_version', 'va_last_error'):\n"
        "        continue\n"
        "    zeros = [None if a is ctypes.c_void_p else (0.0 if a is ctypes.c_float else 0) for a in args]\n"
        "    rc = getattr(lib, name)(*zeros)\n"
        "    if res is ctypes.c_int:\n"
        "        msg = lib.va_last_error().decode()\n"
        "        assert rc < 0, (name, rc)\n"
        "        assert msg.startswith('va_'), (name, msg)\n"
        "    else:\n"
        "        assert rc == 0, (name, rc)      # the size queries answer 0 for an empty problem\n"
        "print('checked', len(_lib.SIGNATURES) - 2)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checked" in r.stdout
