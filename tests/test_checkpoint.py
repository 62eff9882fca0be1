"""CPU: checkpoint files in the reference's dictionary layouts (stage1_surgvu_pretraining.py:498-523,
stage1_vjepa2_finetuning.py:84-128 and 371-390, load_vjepa2_pretrained.py:292-297)."""
import pytest
import torch

from oracle import ref_cpu as R
from vit_amd import checkpoint as C
from vit_amd import surgical_world_model as M
from vit_amd.train_step import Stage1Heads


def tiny_model(depth=1, predictor_depth=2):
    m = M.SurgicalActionConditionedWorldModel(img_size=32, num_frames=4, encoder_depth=depth, encoder_heads=16,
                                              predictor_depth=predictor_depth, predictor_heads=8)
    m.load_state_dict(R.fill_state_dict_(m.state_dict()))
    return m


def test_stage1_layouts(tmp_path):
    model, heads = tiny_model(), Stage1Heads()
    params = list(model.encoder.parameters()) + list(heads.parameters())
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.05)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    C.save_stage1_checkpoint(tmp_path / "checkpoint_epoch_01.pth", 1, model, heads, 0.5, opt, sched)
    C.save_stage1_checkpoint(tmp_path / "best_encoder_surgvu.pth", 1, model, heads, 0.5)
    ep = C.load_checkpoint(tmp_path / "checkpoint_epoch_01.pth")
    best = C.load_checkpoint(tmp_path / "best_encoder_surgvu.pth")
    assert list(ep) == ["epoch", "encoder_state_dict", "temporal_classifier", "contrastive_proj", "denoise_head",
                        "optimizer", "scheduler", "loss"]
    assert list(best) == ["epoch", "encoder_state_dict", "temporal_classifier", "contrastive_proj", "denoise_head", "loss"]
    assert {k: tuple(v.shape) for k, v in best["encoder_state_dict"].items() if "inv_freq" not in k} == R.encoder_param_shapes(1)
    assert list(best["temporal_classifier"]) == ["0.weight", "0.bias", "3.weight", "3.bias"]
    assert list(best["contrastive_proj"]) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    assert list(best["denoise_head"]) == ["0.weight", "0.bias", "1.weight", "1.bias", "3.weight", "3.bias"]
    # the encoder of a fresh model takes the file back exactly
    fresh = tiny_model()
    with torch.no_grad():
        for p in fresh.parameters():
            p.zero_()
    assert fresh.encoder.load_state_dict(best["encoder_state_dict"], strict=False).missing_keys == []
    for (k, a), (_, b) in zip(model.encoder.state_dict().items(), fresh.encoder.state_dict().items()):
        assert torch.equal(a, b), k
    no_aux = Stage1Heads(use_mask_aux=False)
    C.save_stage1_checkpoint(tmp_path / "b.pth", 0, model, no_aux, 1.0)
    assert C.load_checkpoint(tmp_path / "b.pth")["denoise_head"] is None


def test_stage1_files_have_the_manifest_of_the_reference_loops_files(tmp_path, golden_dir):
    """stage1_checkpoint_manifest.json lists the nested keys, shapes and dtypes of the two files the reference's own
    loop wrote (s1:498-523) for a depth-1 model.  The files this package writes -- with torch's AdamW as in the
    reference, and with FusedAdamW in its place -- must have the same structure."""
    import json
    import os
    from conftest import checkpoint_manifest
    want = json.load(open(os.path.join(golden_dir, "stage1_checkpoint_manifest.json")))
    model, heads = tiny_model(predictor_depth=1), Stage1Heads()
    params = list(model.encoder.parameters()) + list(heads.parameters())
    opt = torch.optim.AdamW(params, lr=4e-4, weight_decay=0.1)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1)
    for p in params:
        p.grad = torch.zeros_like(p)
    opt.step()
    sched.step()
    C.save_stage1_checkpoint(tmp_path / "checkpoint_epoch_01.pth", 1, model, heads, 2.3, opt, sched)
    C.save_stage1_checkpoint(tmp_path / "best_encoder_surgvu.pth", 1, model, heads, 2.3)
    for name in ("checkpoint_epoch_01.pth", "best_encoder_surgvu.pth"):
        got = checkpoint_manifest(C.load_checkpoint(tmp_path / name))
        assert got == want[name], name


def test_pretrained_file_has_the_manifest_of_the_reference_writers_file(tmp_path, golden_dir):
    """pretrained_checkpoint_manifest.json: nested keys / shapes / dtypes of the file the reference's own
    `save_adapted_checkpoint` (load_vjepa2_pretrained.py:279-300) wrote for a depth-1 model with a two-block
    predictor.  (The fixture script also had the reference's `load_pretrained_model`, ft:57-138, load a file written by
    this package and compared every tensor.)"""
    import json
    import os
    from conftest import checkpoint_manifest
    want = json.load(open(os.path.join(golden_dir, "pretrained_checkpoint_manifest.json")))
    model = tiny_model(depth=1, predictor_depth=2)
    C.save_pretrained_model(model, tmp_path / "p.pth", {"embed_dim": 1024, "depth": 1, "num_heads": 16}, {"source": "test"})
    assert checkpoint_manifest(C.load_checkpoint(tmp_path / "p.pth")) == want


def test_finetune_layout(tmp_path):
    model, heads = tiny_model(), Stage1Heads()
    opt = torch.optim.AdamW(heads.parameters(), lr=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    C.save_finetune_checkpoint(tmp_path / "best_vjepa2_finetuned.pth", 3, model, heads, 0.25, opt, sched, True)
    ck = C.load_checkpoint(tmp_path / "best_vjepa2_finetuned.pth")
    assert list(ck) == ["epoch", "encoder_state_dict", "full_model_state_dict", "temporal_classifier", "contrastive_proj",
                        "denoise_head", "optimizer", "scheduler", "loss", "config"]
    assert ck["config"] == {"encoder_frozen": True, "encoder_dim": 1024, "pretrained_source": "V-JEPA 2"}
    assert set(ck["full_model_state_dict"]) == set(model.state_dict())


def test_pretrained_round_trip_and_defaults(tmp_path):
    model = tiny_model(depth=2)
    C.save_pretrained_model(model, tmp_path / "p.pth", {"embed_dim": 1024, "depth": 2, "num_heads": 16}, {"src": "t"})
    ck = C.load_checkpoint(tmp_path / "p.pth")
    assert list(ck) == ["model_state_dict", "encoder_state_dict", "vjepa2_config", "metadata"]
    loaded, cfg = C.load_pretrained_model(tmp_path / "p.pth", img_size=32, num_frames=4, predictor_depth=2, device="cpu")
    assert cfg == {"encoder_dim": 1024, "encoder_depth": 2, "encoder_heads": 16}            # ft:132-136
    res = loaded.load_result
    assert res.missing_keys == [] and res.unexpected_keys == []
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b), k
    assert loaded.use_grad_checkpoint is True
    # encoder-only file, different predictor depth: strict=False leaves the predictor at its initial values
    torch.save({"encoder_state_dict": ck["encoder_state_dict"], "vjepa2_config": {"depth": 2}}, tmp_path / "e.pth")
    enc_only, cfg = C.load_pretrained_model(tmp_path / "e.pth", img_size=32, num_frames=4, predictor_depth=1, device="cpu")
    assert cfg["encoder_heads"] == 16 and enc_only.load_result.missing_keys == [] and len(enc_only.predictor_blocks) == 1
    assert torch.equal(enc_only.encoder.blocks[1].mlp[0].weight, model.encoder.blocks[1].mlp[0].weight)
    # the config defaults are the V-JEPA2 ViT-L numbers
    assert C.VJEPA2_DEFAULTS == {"embed_dim": 1024, "depth": 24, "num_heads": 16}
    torch.save({"something": 1}, tmp_path / "bad.pth")
    with pytest.raises(ValueError):
        C.load_pretrained_model(tmp_path / "bad.pth", img_size=32, num_frames=4, predictor_depth=1, device="cpu")
