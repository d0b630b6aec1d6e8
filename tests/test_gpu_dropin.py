"""Drop-in surface on the GPU: the reference drivers' call sequence (trace/eval/evaluate.py:241-243,315-410) run
against trace_amd with a synthetic tiny checkpoint."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
if not torch.cuda.is_available():
    pytest.skip("needs a HIP device", allow_module_level=True)

from oracle import trace_oracle as O  # noqa: E402  (checker only)
from trace_amd import config as tcfg, synth  # noqa: E402
from trace_amd.constants import DEFAULT_MMODAL_TOKEN  # noqa: E402
from trace_amd.conversation import conv_templates  # noqa: E402
from trace_amd.mm_utils import get_model_name_from_path, process_video, tokenizer_MMODAL_token_all  # noqa: E402
from trace_amd.model.builder import load_pretrained_model, save_synthetic_checkpoint  # noqa: E402


@pytest.fixture(scope="module")
def loaded(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("ckpt") / "trace-tiny")
    cfg = tcfg.tiny(num_frames=4)
    save_synthetic_checkpoint(path, cfg)
    tok, model, proc, ctx_len = load_pretrained_model(path, None, get_model_name_from_path(path), max_batch=2, max_new_tokens=64)
    return cfg, tok, model, proc, ctx_len


def _driver_inputs(cfg, tok, proc, seed=0):
    raw = np.random.RandomState(seed).randint(0, 255, size=(40, 48, 64, 3), dtype=np.uint8)
    tensor, ts = process_video(raw, proc, "pad", 4, fps=8.0)
    assert tensor.shape == (4, 3, cfg.vision_image_size, cfg.vision_image_size)
    conv = conv_templates["llama_2"].copy()
    conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\nfind events")
    conv.append_message(conv.roles[1], None)
    prompt = conv.get_prompt() + "<sync>"
    ids = tokenizer_MMODAL_token_all(prompt, tok, return_tensors="pt")
    vp = int(torch.nonzero(ids == -201)[0])
    ids = torch.cat([ids[:1], ids[vp - 20: vp + 20], ids[-3:]])          # keep the byte-level prompt short
    assert (ids == -201).sum() == 1 and ids[-1] == -205
    return tensor, ts, ids


def test_driver_loop_matches_oracle(loaded):
    cfg, tok, model, proc, ctx_len = loaded
    assert ctx_len == cfg.max_sequence_length and proc is not None
    tensor, ts, ids = _driver_inputs(cfg, tok, proc)
    heads = [1]
    out = model.generate(ids.unsqueeze(0).to("cuda"), attention_mask=None, images_or_videos=[tensor.to(torch.float16).to("cuda")],
                         modal_list=["video"], do_sample=False, temperature=0.0, max_new_tokens=12, use_cache=True,
                         pad_token_id=tok.eos_token_id, video_timestamps=[ts], heads=heads)
    assert out.shape[0] == 1 and out.dtype == torch.long
    ora = O.Oracle(cfg, synth.state_dict(cfg), emulate_bf16=True)
    fr = tensor.to(torch.float16).to(torch.bfloat16).float()
    ref, lg = ora.generate(ids, fr, ts, head=1, max_new_tokens=12, eos_token_id=cfg.eos_token_id, return_logits=True)
    srt = torch.sort(torch.where(torch.isfinite(lg), lg, torch.full_like(lg, -1e30)), dim=-1, descending=True).values
    for i, (a, b) in enumerate(zip(out[0].tolist(), ref)):
        if (srt[i, 0] - srt[i, 1]) < 0.1:
            break
        assert a == b, f"step {i}"
    # the parser of the drivers only needs decode(int-or-tensor) on the number tokenizers
    tt = model.get_model().time_tokenizer
    assert tt.decode(out[0][0] - (cfg.vocab_size + 1)) in "<sync><sep>0123456789."
    assert heads[0] in (0, 1, 2)


def test_sampling_and_asserts(loaded):
    cfg, tok, model, proc, _ = loaded
    tensor, ts, ids = _driver_inputs(cfg, tok, proc, seed=1)
    torch.manual_seed(0)
    out = model.generate(ids.unsqueeze(0), images_or_videos=[tensor], modal_list=["video"], do_sample=True, temperature=0.2,
                         max_new_tokens=8, video_timestamps=[ts], heads=[1])
    assert out.shape == (1, 8) or out.shape[1] <= 8
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    assert V < int(out[0, 0]) <= V + Tv          # generation starts in the time head
    with pytest.raises(NotImplementedError):
        model.generate(ids.unsqueeze(0), images_or_videos=[tensor], inputs_embeds=torch.zeros(1), video_timestamps=[ts], heads=[1])
    with pytest.raises(Exception, match="only have one video"):
        bad = torch.cat([ids, torch.tensor([-201])])
        model.generate(bad.unsqueeze(0), images_or_videos=[tensor], modal_list=["video"], video_timestamps=[ts], heads=[1])


def test_safetensors_checkpoint_loader(tmp_path_factory, loaded):
    """HF-format checkpoint path: weights saved as sharded safetensors under the reference's
    state-dict names — CLIP keys in the transformers-5 layout (no `.vision_model`) in one shard to exercise both
    spellings — must load to exactly the same model as the synthetic path."""
    from safetensors.torch import save_file
    cfg, tok, model, proc, _ = loaded
    path = str(tmp_path_factory.mktemp("ckpt_st") / "trace-tiny-st")
    os.makedirs(path)
    cfg.save_pretrained(path)
    sd = synth.state_dict(cfg)
    a, b = {}, {}
    for i, (k, v) in enumerate(sd.items()):
        if "vision_tower.vision_tower.vision_model." in k and "encoder.layers.1." in k:
            k = k.replace("vision_tower.vision_tower.vision_model.", "vision_tower.vision_tower.")
        (a if i % 2 else b)[k] = v.contiguous()
    save_file(a, os.path.join(path, "model-00001-of-00002.safetensors"))
    save_file(b, os.path.join(path, "model-00002-of-00002.safetensors"))
    with pytest.warns(UserWarning, match="byte-level"):
        tok2, model2, proc2, _ = load_pretrained_model(path, None, get_model_name_from_path(path), max_batch=1, max_new_tokens=32)
    tensor, ts, ids = _driver_inputs(cfg, tok, proc)
    kw = dict(images_or_videos=[tensor], modal_list=["video"], do_sample=False, max_new_tokens=10, video_timestamps=[ts])
    o1 = model.generate(ids.unsqueeze(0), heads=[1], **kw)
    o2 = model2.generate(ids.unsqueeze(0), heads=[1], **kw)
    assert torch.equal(o1, o2)
    model2.engine.close()


def test_batched_generate_and_device_preprocess(loaded):
    """the drivers' API with B > 1 (`images_or_videos=[v1, v2]`, `heads=[1, 1]`, equal prompt lengths — the only
    case the reference's own forward supports, trace_arch.py:502-517) gives each row what the single-video call gives, pads
    with pad_token_id and mutates `heads` per row; and frames preprocessed on the device (process_video(engine=model)) decode
    to the same ids as frames from the host PIL path."""
    cfg, tok, model, proc, _ = loaded
    t1, ts1, ids = _driver_inputs(cfg, tok, proc, seed=1)
    t2, ts2, _ = _driver_inputs(cfg, tok, proc, seed=2)
    kw = dict(modal_list=["video"], do_sample=False, max_new_tokens=10, use_cache=True, pad_token_id=tok.eos_token_id)
    o1 = model.generate(ids.unsqueeze(0), images_or_videos=[t1], video_timestamps=[ts1], heads=[1], **kw)
    o2 = model.generate(ids.unsqueeze(0), images_or_videos=[t2], video_timestamps=[ts2], heads=[1], **kw)
    heads = [1, 1]
    kw["modal_list"] = ["video", "video"]
    ob = model.generate(torch.stack([ids, ids]), images_or_videos=[t1, t2], video_timestamps=[ts1, ts2], heads=heads, **kw)
    assert ob.shape[0] == 2 and ob.dtype == torch.long
    assert ob[0, : o1.shape[1]].tolist() == o1[0].tolist() and ob[1, : o2.shape[1]].tolist() == o2[0].tolist()
    assert all(h in (0, 1, 2) for h in heads)
    # device preprocessing inside the driver loop
    raw = np.random.RandomState(1).randint(0, 255, size=(40, 48, 64, 3), dtype=np.uint8)
    dev, ts_d = process_video(raw, proc, "pad", 4, fps=8.0, engine=model)
    assert ts_d == ts1 and dev.is_cuda and torch.equal(dev.cpu(), t1.to(torch.bfloat16))
    kw["modal_list"] = ["video"]
    od = model.generate(ids.unsqueeze(0), images_or_videos=[dev], video_timestamps=[ts_d], heads=[1], **kw)
    assert od.tolist() == o1.tolist()


def test_evaluate_videos_equals_the_driver_loop(loaded):
    """trace_amd.evaluate.evaluate_videos (batched, device preprocessing) returns, per video and in input order, exactly
    what the reference's one-video-at-a-time loop (evaluate.py:298-417) produces through model.generate + its parser."""
    from trace_amd import evaluate as ev
    cfg, tok, model, proc, _ = loaded
    rng = np.random.RandomState(11)
    items = [{"id": f"v{i}", "video": rng.randint(0, 255, size=(30, 48, 64 if i != 2 else 40, 3), dtype=np.uint8), "fps": 10.0}
             for i in range(3)]
    prompt = "find events"

    orig = ev.build_prompt_ids

    def short_ids(q):                      # the byte-level stand-in tokenizer makes the llama_2 system prompt very long
        ids = orig(q, tok)
        vp = int(torch.nonzero(ids == -201)[0])
        return torch.cat([ids[:1], ids[vp - 20: vp + 20], ids[-3:]])
    ev.build_prompt_ids = lambda q, t, conv_mode="llama_2": short_ids(q)
    try:
        res = ev.evaluate_videos(model, tok, proc, items, prompt, num_frames=4, max_new_tokens=10, batch_size=2)
        # two banks of KV slots (max_batch 2, one video per chunk): the chunks go through the two-stage pipeline — the same ids
        res_p = ev.evaluate_videos(model, tok, proc, items, prompt, num_frames=4, max_new_tokens=10, batch_size=1, pipeline=True)
        assert [r["output_ids"] for r in res_p] == [r["output_ids"] for r in res]
    finally:
        ev.build_prompt_ids = orig
    assert [r["id"] for r in res] == ["v0", "v1", "v2"]
    for it, r in zip(items, res):
        tensor, ts = process_video(it["video"], proc, "pad", 4, fps=it["fps"])
        out = model.generate(short_ids(prompt).unsqueeze(0), images_or_videos=[tensor], modal_list=["video"], do_sample=False,
                             max_new_tokens=10, use_cache=True, pad_token_id=tok.eos_token_id, video_timestamps=[ts], heads=[1])
        n = len(r["output_ids"])
        assert out[0, :n].tolist() == r["output_ids"]
        try:
            ref = ev.parse_output_ids(out[0, :n].tolist(), tok, model, ev.stop_string())
        except ValueError:             # random weights can emit malformed numbers; the batched driver records that as an error
            assert "error" in r
            continue
        assert (r["timestamps"], r["scores"], r["captions"]) == (ref["timestamps"], ref["scores"], ref["captions"])


def test_driver_loop_through_sentencepiece_tokenizer(tmp_path_factory, golden_dir):
    """a checkpoint directory as the reference ships one — config.json, safetensors shards under the reference's
    state-dict names, sentencepiece tokenizer.model + tokenizer_config.json — loads through AutoTokenizer (builder.py:113) and the
    drivers' loop (conversation -> tokenizer_MMODAL_token_all -> generate -> batch_decode / KeywordsStoppingCriteria) runs on it."""
    import dataclasses
    import shutil
    from safetensors.torch import save_file
    from trace_amd.mm_utils import KeywordsStoppingCriteria
    cfg = dataclasses.replace(tcfg.tiny(num_frames=4), vocab_size=512)          # the committed sentencepiece model has 512 pieces
    path = str(tmp_path_factory.mktemp("ckpt_sp") / "trace-tiny-sp")
    os.makedirs(path)
    cfg.save_pretrained(path)
    save_file({k: v.contiguous() for k, v in synth.state_dict(cfg).items()}, os.path.join(path, "model.safetensors"))
    for f in ("tokenizer.model", "tokenizer_config.json"):
        shutil.copy(os.path.join(golden_dir, "sp_tiny", f), path)
    tok, model, proc, _ = load_pretrained_model(path, None, get_model_name_from_path(path), max_batch=1, max_new_tokens=32)
    assert type(tok).__name__.startswith("Llama") and len(tok) == cfg.vocab_size
    raw = np.random.RandomState(4).randint(0, 255, size=(40, 48, 64, 3), dtype=np.uint8)
    tensor, ts = process_video(raw, proc, "pad", 4, fps=8.0)
    conv = conv_templates["llama_2"].copy()
    conv.append_message(conv.roles[0], DEFAULT_MMODAL_TOKEN["VIDEO"] + "\nfind events")
    conv.append_message(conv.roles[1], None)
    ids = tokenizer_MMODAL_token_all(conv.get_prompt() + "<sync>", tok, return_tensors="pt")
    assert (ids == -201).sum() == 1 and ids[-1] == -205 and ids[0] == tok.bos_token_id and int(ids.max()) < cfg.vocab_size
    heads = [1]
    out = model.generate(ids.unsqueeze(0).to("cuda"), images_or_videos=[tensor.to("cuda")], modal_list=["video"], do_sample=False,
                         max_new_tokens=12, use_cache=True, pad_token_id=tok.eos_token_id, video_timestamps=[ts], heads=heads)
    ora = O.Oracle(cfg, synth.state_dict(cfg), emulate_bf16=True)
    ref, lg = ora.generate(ids, tensor.to(torch.bfloat16).float(), ts, head=1, max_new_tokens=12, eos_token_id=cfg.eos_token_id,
                           return_logits=True)
    srt = torch.sort(torch.where(torch.isfinite(lg), lg, torch.full_like(lg, -1e30)), dim=-1, descending=True).values
    for i, (a, b) in enumerate(zip(out[0].tolist(), ref)):
        if (srt[i, 0] - srt[i, 1]) < 0.1:
            break
        assert a == b, f"step {i}"
    text_ids = [t for t in out[0].tolist() if t < cfg.vocab_size]
    assert isinstance(tok.batch_decode([text_ids], skip_special_tokens=True)[0], str)
    sc = KeywordsStoppingCriteria(["</s>"], tok, ids.unsqueeze(0))
    out2 = model.generate(ids.unsqueeze(0), images_or_videos=[tensor], modal_list=["video"], do_sample=False, max_new_tokens=12,
                          stopping_criteria=[sc], video_timestamps=[ts], heads=[1])
    n = out2.shape[1]
    assert out2[0].tolist() == out[0, :n].tolist()              # the stepwise (stopping-criteria) path emits the same greedy ids
    model.engine.close()


def test_forward_contract_prefill_and_decode_forms(loaded):
    """forward() as the reference defines it (trace_mistral.py:114-264): prefill form -> logits for EVERY position [B, L, V'] with the
    head mask, decode form (input_ids [B,1] + past_key_values) -> [B, 1, V']; both against the bf16-emulating oracle, and the
    last prefill position / the decode steps equal what generate()'s own loop computes."""
    cfg, tok, model, proc, _ = loaded
    tensor, ts, ids = _driver_inputs(cfg, tok, proc, seed=3)
    ora = O.Oracle(cfg, synth.state_dict(cfg), emulate_bf16=True)
    fr = tensor.to(torch.bfloat16)
    out = model.forward(input_ids=ids.unsqueeze(0), images=[[fr], ["video"]], video_timestamps=[ts], heads=[1], use_cache=True)
    NV = cfg.total_vocab
    vid = ora.encode_video(fr.float(), ts)
    emb = ora.splice(ids, vid)
    hid, kv = ora.llm_forward(emb)
    ref = ora.logits(hid, 1)
    L = emb.shape[0]
    assert out.logits.shape == (1, L, NV) and out.logits.dtype == torch.float32
    got = out.logits[0].cpu()
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref))
    fin = torch.isfinite(ref)
    assert (got[fin] - ref[fin]).abs().max().item() < 0.08
    # heads=None: text | sync logits only, unmasked
    out0 = model.forward(input_ids=ids.unsqueeze(0), images=[[fr], ["video"]], video_timestamps=[ts])
    assert out0.logits.shape == (1, L, cfg.vocab_size + 1) and torch.isfinite(out0.logits).all()
    # decode form: feed the arg-max of the last position, twice; heads follow the swap rule
    out = model.forward(input_ids=ids.unsqueeze(0), images=[[fr], ["video"]], video_timestamps=[ts], heads=[1], use_cache=True)
    heads, lg = [1], out.logits[0, -1]
    pkv = out.past_key_values
    for _ in range(3):
        t = int(torch.argmax(lg))
        heads[0] = model.swap_tokens.get(t, heads[0])
        o = model.forward(input_ids=torch.tensor([[t]]), past_key_values=pkv, heads=heads, use_cache=True)
        assert o.logits.shape == (1, 1, NV)
        hid, kv = ora.llm_forward(ora.decode_embed(t)[None], kv)
        r = ora.logits(hid, heads[0])[0]
        lg = o.logits[0, 0].cpu()
        f = torch.isfinite(r)
        assert torch.equal(torch.isfinite(lg), f) and (lg[f] - r[f]).abs().max().item() < 0.08
        pkv = o.past_key_values
    with pytest.raises(ValueError, match="disagrees"):
        model.forward(input_ids=torch.tensor([[5]]), past_key_values=pkv, heads=[(heads[0] + 1) % 3])
    # generate() after forward() works (host mode is reset)
    g = model.generate(ids.unsqueeze(0), images_or_videos=[tensor], modal_list=["video"], do_sample=False, max_new_tokens=4,
                       video_timestamps=[ts], heads=[1])
    assert g.shape[0] == 1
