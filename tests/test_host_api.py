"""Host-side mirror of the reference's Python harness (trace_amd.{constants,conversation,mm_utils,model.encoders})
against vectors captured from the reference (tests/golden/host_functions.json).  CPU only."""
import json
import os
import types

import numpy as np
import pytest
import torch
from PIL import Image

from trace_amd import constants, conversation, mm_utils
from trace_amd.model.encoders import NumberTokenizer, ScoreTower, TimeTower


@pytest.fixture(scope="module")
def G(golden_dir):
    return json.load(open(os.path.join(golden_dir, "host_functions.json")))


def test_constants(G):
    for k, v in G["constants"].items():
        assert getattr(constants, k) == v, k


def test_time_score_towers(G):
    tt, st = TimeTower(), ScoreTower()
    for c in G["time_encode"]:
        assert tt.encode(c["in"]).tolist() == c["out"], c
    for c in G["score_encode"]:
        assert st.encode(c["in"]).tolist() == c["out"], c
    tok = NumberTokenizer()
    for c in G["time_decode"]:
        assert tok.decode(c["in"]) == c["out"]
    for c in G["score_decode"]:
        assert tok.decode(torch.tensor(c["in"])) == c["out"]        # drivers pass 0-d tensors (evaluate.py:395)
    assert tok.get_vocab() == G["time_vocab"]


def test_llama2_prompts(G):
    for name, c in G["llama2_prompts"].items():
        if name == "_multi_turn":
            continue
        conv = conversation.conv_templates["llama_2"].copy()
        conv.append_message(conv.roles[0], "<video>\n" + c["question"])
        conv.append_message(conv.roles[1], None)
        assert conv.get_prompt() + "<sync>" == c["prompt"], name
    conv = conversation.conv_templates["llama_2"].copy()
    conv.append_message(conv.roles[0], "<video>\nhello")
    conv.append_message(conv.roles[1], "an answer")
    conv.append_message(conv.roles[0], "second turn")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == G["llama2_prompts"]["_multi_turn"]["prompt"]
    assert [conv.sep, conv.sep2] == G["llama2_sep"]
    # copy() must not alias the template's message list
    assert conversation.conv_templates["llama_2"].messages == []


class FakeTok:
    bos_token_id = 1

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[1] + [10 + len(w) for w in text.split()])


def test_tokenizer_mmodal(G):
    for c in G["tokenizer_MMODAL_token_all"]:
        assert mm_utils.tokenizer_MMODAL_token_all(c["in"], FakeTok(), return_tensors="pt").tolist() == c["out"], c["in"]
    for c in G["tokenizer_MMODAL_token_video"]:
        assert mm_utils.tokenizer_MMODAL_token(c["in"], FakeTok(), -201, return_tensors="pt").tolist() == c["out"], c["in"]
    for c in G["get_model_name_from_path"]:
        assert mm_utils.get_model_name_from_path(c["in"]) == c["out"]


def test_frame_sampling_and_timestamps(G):
    for c in G["frame_sample_uniform"]:
        idx, ts = mm_utils.sample_indices_and_timestamps(c["duration"], c["fps"], c["num_frames"])
        assert idx.tolist() == c["indices"]
        assert ts == c["timestamps"]


def test_expand2square(G):
    for c in G["expand2square"]:
        out = mm_utils.expand2square(Image.fromarray(np.array(c["in"], dtype=np.uint8)), tuple(c["bg"]))
        assert np.array(out).tolist() == c["out"]


class FakeProcessor:
    image_mean = [0.48145466, 0.4578275, 0.40821073]

    def preprocess(self, images, return_tensors="pt"):
        arr = np.stack([np.asarray(im.resize((28, 28)), dtype=np.float32) / 255.0 for im in images])
        return {"pixel_values": torch.from_numpy(arr).permute(0, 3, 1, 2)}


def test_process_video_from_frames():
    frames = np.random.RandomState(0).randint(0, 255, size=(50, 20, 30, 3), dtype=np.uint8)
    vid, ts = mm_utils.process_video(frames, FakeProcessor(), "pad", num_frames=8, fps=10.0)
    assert vid.shape == (8, 3, 28, 28)
    assert ts == [[float(i / 10.0)] for i in np.linspace(0, 49, 8, dtype=int)]
    with pytest.raises(ImportError, match="too long"):
        mm_utils.process_video(np.zeros((20, 4, 4, 3), np.uint8), FakeProcessor(), None, num_frames=4, fps=0.001)


def test_compat_aliases():
    from trace_amd import compat
    compat.install()
    from Trace.trace.conversation import conv_templates, SeparatorStyle  # noqa: F401
    from Trace.trace.constants import DEFAULT_MMODAL_TOKEN, MMODAL_TOKEN_INDEX  # noqa: F401
    from Trace.trace.mm_utils import get_model_name_from_path, tokenizer_MMODAL_token_all, process_video, KeywordsStoppingCriteria  # noqa: F401
    assert MMODAL_TOKEN_INDEX["SYNC"] == -205


def test_bench_dvc_schedule_walks_the_heads():
    """bench.py's forced feed schedule: per event 14 time-head, 4 score-head and 33 text-head steps, every fed
    token inside the vocabulary range of the head that is active when it is produced (trace_mistral.py:86-88,244-252)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracle import trace_oracle as O
    from trace_amd import config as tcfg
    cfg = tcfg.trace_7b(128)
    ids = bench.dvc_schedule(cfg, 256, seed=3)
    assert len(ids) == 256
    head, counts = 1, {0: 0, 1: 0, 2: 0}
    for t in ids:
        lo, hi = O.head_range(cfg, head)
        assert lo <= t < hi, (t, head)
        counts[head] += 1
        head = O.swap_head(cfg, t, head)
    assert counts[1] == 14 * 5 + 1 and counts[2] == 4 * 5 and counts[0] == 33 * 5, counts


def test_evaluate_parser_and_prompt():
    """trace_amd.evaluate: the drivers' id-stream parser (evaluate.py:360-411) against the oracle's restatement on a stream
    that visits all three heads, and the prompt builder against the captured llama_2 + <sync> placeholder layout."""
    from types import SimpleNamespace
    from oracle import trace_oracle as O
    from trace_amd import config as tcfg, evaluate as ev
    from trace_amd.model.encoders import NumberTokenizer
    cfg = tcfg.trace_7b(8)
    V, Tv = cfg.vocab_size, cfg.time_vocab_size
    meta = SimpleNamespace(time_tokenizer=NumberTokenizer(), score_tokenizer=NumberTokenizer())
    model = SimpleNamespace(config=cfg, get_model=lambda: meta)
    tok = SimpleNamespace(decode=lambda ids, skip_special_tokens=True: " ".join(str(i) for i in ids))
    t = lambda s: [V + 1 + O.NUM_VOCAB[ch] for ch in s]
    sc = lambda s: [V + Tv + 1 + O.NUM_VOCAB[ch] for ch in s]
    ids = (t("0012.5") + [V + 2] + t("0030.0") + [V + 1] + sc("4.5") + [V + Tv + 1] + [11, 12, 13, V]
           + t("0040.0") + [V + 2] + t("0055.5") + [V + 1] + sc("3.0") + [V + Tv + 1] + [21, 22])
    got = ev.parse_output_ids(ids, tok, model)
    ref = O.parse_output_ids(cfg, ids)
    assert got["timestamps"] == ref["timestamps"] == [[12.5, 30.0], [40.0, 55.5]]
    assert got["scores"] == ref["scores"] == [[4.5], [3.0]]
    assert got["captions"] == ["11 12 13", "21 22"] and ref["captions"][0] == [11, 12, 13]
    # stop string inside a flushed caption ends the parse (evaluate.py:380-381)
    tok2 = SimpleNamespace(decode=lambda ids, skip_special_tokens=True: "done </s>")
    assert ev.parse_output_ids([5, V, 6, V], tok2, model, stop_str="</s>")["captions"] == ["done </s>"]
    # prompt: BOS ... <video> ... <sync> last
    from trace_amd.model.builder import ByteTokenizer
    p = ev.build_prompt_ids("find events", ByteTokenizer(cfg.vocab_size))
    assert int((p == -201).sum()) == 1 and int(p[-1]) == -205 and ev.stop_string() == "</s>"


def test_checkpoint_iterator_layouts(tmp_path):
    """SURVEY §8f-2 host logic (no GPU): a TRACE checkpoint directory whose shards are mixed safetensors-less `.bin` files with
    the serialised rotary buffers older transformers wrote, NO vision tower inside, and `mm_vision_tower` pointing at a local
    CLIPModel directory (vision_model.* beside text_model.* / logit_scale, geometry in its config.json) must yield exactly the
    tensors of the path under the reference's state-dict names."""
    import dataclasses, json
    from trace_amd import config as tcfg, synth
    from trace_amd.model import builder
    cfg = tcfg.tiny()
    clip_dir = tmp_path / "tiny-clip"
    clip_dir.mkdir()
    sd = synth.state_dict(cfg)
    vis = {k[len("model.vision_tower.vision_tower."):]: v for k, v in sd.items() if ".vision_tower." in k}
    vis["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)
    vis["logit_scale"] = torch.tensor(1.0)
    vis["vision_model.embeddings.position_ids"] = torch.arange(17).unsqueeze(0)
    torch.save(vis, clip_dir / "pytorch_model.bin")
    json.dump({"model_type": "clip", "vision_config": {
        "hidden_size": cfg.vision_hidden_size, "intermediate_size": cfg.vision_intermediate_size, "num_hidden_layers": cfg.vision_num_layers,
        "num_attention_heads": cfg.vision_num_heads, "image_size": cfg.vision_image_size, "patch_size": cfg.vision_patch_size,
        "layer_norm_eps": cfg.vision_layer_norm_eps}}, open(clip_dir / "config.json", "w"))
    ckpt = tmp_path / "trace-tiny"
    ckpt.mkdir()
    d = dataclasses.replace(cfg, mm_vision_tower=str(clip_dir)).to_dict()
    for k in list(d):
        if k.startswith("vision_") and k != "vision_layers_used":
            d.pop(k)                                  # a real TRACE config.json does not spell the CLIP geometry out
    json.dump(d, open(ckpt / "config.json", "w"))
    rest = [(k, v) for k, v in sd.items() if ".vision_tower." not in k]
    a = dict(rest[0::2])
    b = dict(rest[1::2])
    b["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(64)
    torch.save(a, ckpt / "pytorch_model-00001-of-00002.bin")
    torch.save(b, ckpt / "pytorch_model-00002-of-00002.bin")
    cfg2 = tcfg.TraceConfig.from_pretrained(str(ckpt))
    assert dataclasses.replace(cfg2, mm_vision_tower=cfg.mm_vision_tower) == cfg
    got = dict(builder._iter_checkpoint(str(ckpt), cfg2))
    assert set(got) == set(sd)
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # a checkpoint that does carry the tower wins over the CLIP directory
    torch.save({k: v + 1 for k, v in sd.items() if ".vision_tower." in k}, ckpt / "pytorch_model-00003-of-00003.bin")
    got = dict(builder._iter_checkpoint(str(ckpt), cfg2))
    k0 = synth.VIS + "pre_layrnorm.weight"
    assert torch.equal(got[k0], sd[k0] + 1)
    # no tower anywhere -> a clear error, not a hub download
    (ckpt / "pytorch_model-00003-of-00003.bin").unlink()
    bad = dataclasses.replace(cfg2, mm_vision_tower="openai/clip-vit-large-patch14-336")
    with pytest.raises(FileNotFoundError, match="no hub access"):
        dict(builder._iter_checkpoint(str(ckpt), bad))


class _IndexCodedReader:
    """decord-like reader whose frame k is a 4x6 image filled with k % 251 (what oracle/make_goldens.py fed the reference)"""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def get_batch(self, ids):
        arr = np.stack([np.full((4, 6, 3), int(i) % 251, dtype=np.uint8) for i in ids])
        return types.SimpleNamespace(numpy=lambda: arr)


class _ReportingProcessor:
    image_mean = [0.48145466, 0.4578275, 0.40821073]

    def preprocess(self, images, return_tensors="pt"):
        return {"pixel_values": torch.tensor([[int(np.asarray(im)[im.size[1] // 2, im.size[0] // 2, 0]), im.size[0], im.size[1]]
                                              for im in images])}


def test_process_video_against_reference_runs(G, tmp_path):
    """process_video end to end (sampling scheme, MAX_FRAMES cap, GIF de-duplication at 10 fps, photo grid, padding, timestamp
    guards) against what the reference's own process_video returned for the same index-coded clips (captured through fake
    readers by oracle/make_goldens.py): same frames in the same order, same sizes after padding, same timestamps, same errors."""
    import random
    for c in G["process_video"]:
        if c["kind"] == "gif":
            path = str(tmp_path / f"clip{c['duration']}.gif")
            frames = [Image.fromarray(np.full((4, 6), 10 * (k % 25), dtype=np.uint8), mode="L") for k in range(c["duration"])]
            frames[0].save(path, save_all=True, append_images=frames[1:], duration=100, loop=0)
            v, ts = mm_utils.process_video(path, _ReportingProcessor(), aspect_ratio="pad", num_frames=c["num_frames"])
        else:
            random.seed(1234)
            call = lambda: mm_utils.process_video(_IndexCodedReader(c["duration"]), _ReportingProcessor(), aspect_ratio=c["aspect"],
                                                  num_frames=c["num_frames"], image_grid=c["image_grid"],
                                                  sample_scheme=c["scheme"], fps=c["fps"])
            if "error" in c:
                kind, msg = c["error"].split(": ", 1)
                with pytest.raises(ImportError) as ei:
                    call()
                assert kind == "ImportError" and str(ei.value) == msg
                continue
            v, ts = call()
        assert v.tolist() == c["picked"], c
        assert ts == c["timestamps"], c


def test_create_photo_grid(G):
    for c in G["create_photo_grid"]:
        arr = np.arange(c["t"] * 2 * 3 * 3, dtype=np.uint8).reshape(c["t"], 2, 3, 3)
        assert mm_utils.create_photo_grid(arr, c["rows"], c["cols"]).tolist() == c["out"], (c["t"], c["rows"], c["cols"])


def test_tower_call_size_fills_whole_gemm_rounds():
    """Host logic of the frame-stream tower: the default frames-per-call makes the token rows an (almost) exact multiple of
    64 row tiles, so the 4 / 12 / 16-column-tile ViT GEMMs run in whole rounds of the 256 CUs, and keeps the fc1 input panel
    inside the 256 MB Infinity Cache."""
    from trace_amd import config as tcfg
    from trace_amd.engine import TraceEngine
    cfg = tcfg.trace_7b(128)
    F = TraceEngine.full_round_frames(cfg)
    assert F == 170
    tiles = -(-F * cfg.vision_tokens // 256)
    assert tiles == 384 and all((tiles * n) % 256 == 0 for n in (4, 12, 16))
    assert F * cfg.vision_tokens * cfg.vision_hidden_size * 2 < 256 << 20
    assert TraceEngine.full_round_frames(tcfg.tiny()) == 256          # capped


def test_sentencepiece_tokenizer_hookup(golden_dir):
    """a model directory with sentencepiece files goes through AutoTokenizer (trace/model/builder.py:113) and
    tokenizer_MMODAL_token_all / tokenizer_MMODAL_token on it give what the REFERENCE's functions gave with the same tokenizer
    (tests/golden/tokenizer_sp.json, captured by oracle/make_goldens.py from the imported reference)."""
    import json
    import os
    import torch
    from trace_amd import config as tcfg, mm_utils
    from trace_amd.model import builder
    G = json.load(open(os.path.join(golden_dir, "tokenizer_sp.json")))
    tok = builder.load_tokenizer(os.path.join(golden_dir, "sp_tiny"), tcfg.tiny())
    assert type(tok).__name__.startswith("Llama") and len(tok) == G["vocab_size"] == 512
    assert (tok.bos_token_id, tok.eos_token_id) == (G["bos"], G["eos"])
    for c in G["all"]:
        assert mm_utils.tokenizer_MMODAL_token_all(c["in"], tok, return_tensors="pt").tolist() == c["out"], c["in"][:40]
    for c in G["video"]:
        assert mm_utils.tokenizer_MMODAL_token(c["in"], tok, -201, return_tensors="pt").tolist() == c["out"], c["in"][:40]
    for c in G["decode"]:
        assert tok.batch_decode([c["in"]], skip_special_tokens=True)[0] == c["out"]
    # the drivers' stopping criterion on the real tokenizer: fires on the template's sep2 (evaluate.py:337-341)
    ids = torch.tensor([G["all"][0]["out"][:8]])
    sc = mm_utils.KeywordsStoppingCriteria(["</s>"], tok, ids)
    assert sc(torch.cat([ids, torch.tensor([[40, 41, tok.eos_token_id]])], 1), None) is True
    assert sc(torch.cat([ids, torch.tensor([[40, 41, 42]])], 1), None) is False
