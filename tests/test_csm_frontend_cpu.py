"""CSM string front end (reference sesame.py:427-449 load_llama3_tokenizer, :484-541 _tokenize_text_segment / _tokenize_audio /
_tokenize_segment, :727-757 prompt layout) on a LOCAL byte-level BPE tokenizer built in tmp_path with the `tokenizers` package -- the real
unsloth/Llama-3.2-1B files cannot be fetched offline; the mechanics under test (BOS / EOS template, `[speaker]` prefix, frame layout) do not
depend on the vocabulary.  CPU only: the Mimi codes of the reference clip are injected."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import mlx_audio_amd.params as P  # noqa: E402
from mlx_audio_amd.sesame import Model, Segment, _load_llama3_tokenizer  # noqa: E402

BOS, EOS = "<|begin_of_text|>", "<|end_of_text|>"


@pytest.fixture()
def tok_dir(tmp_path):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    corpus = ["[0]hello there, how are you today?", "[1]fine thanks. and you?", "the quick brown fox jumps over the lazy dog"] * 4
    tk.train_from_iterator(corpus, trainers.BpeTrainer(vocab_size=300, special_tokens=[BOS, EOS], initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    d = tmp_path / "llama3-tokenizer"
    d.mkdir()
    tk.save(str(d / "tokenizer.json"))
    json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "bos_token": BOS, "eos_token": EOS, "model_max_length": 4096}, open(d / "tokenizer_config.json", "w"))
    return str(d)


def test_load_llama3_tokenizer_wraps_text_in_bos_and_eos(tok_dir):
    tok = _load_llama3_tokenizer(tok_dir)
    bos, eos = tok.bos_token_id, tok.eos_token_id
    assert tok.bos_token == BOS and tok.eos_token == EOS and bos != eos
    ids = tok.encode("[0]hello there")
    assert ids[0] == bos and ids[-1] == eos and bos not in ids[1:-1] and eos not in ids[1:-1]  # sesame.py:431-437: "{bos}:0 $A:0 {eos}:0"
    assert tok.decode(ids[1:-1]) == "[0]hello there"
    pair = tok("[0]a", "[1]b")["input_ids"]  # the pair template: both halves wrapped
    assert pair.count(bos) == 2 and pair.count(eos) == 2
    with pytest.raises(Exception):
        _load_llama3_tokenizer(os.path.join(tok_dir, "missing"))  # local_files_only: never a download


def test_string_prompts_give_the_reference_frame_layout(tok_dir):
    cfg = dict(P.csm_tiny_config(), text_tokenizer=tok_dir)
    m = Model(cfg)  # no weights, no codec: host-side prompt building only
    tok, n = m._text_tokenizer, cfg["audio_num_codebooks"]
    # text segment (sesame.py:484-498): ids of "[speaker]text" incl. BOS / EOS in the LAST column, mask there only
    f, k = m._tokenize_text_segment("hello there", 1)
    ids = tok.encode("[1]hello there")
    assert f.shape == k.shape == (len(ids), n + 1) and f[:, -1].tolist() == ids and not f[:, :-1].any()
    assert k[:, -1].all() and not k[:, :-1].any()
    # a context segment with audio (sesame.py:500-541): [text rows | audio rows + one all-zero EOS frame], audio in columns 0..n-1
    rng = np.random.default_rng(0)
    clip = (0.1 * rng.standard_normal(1920 * 3)).astype(np.float32)
    codes = rng.integers(1, cfg["audio_vocab_size"], (n, 3))
    seg = Segment(speaker=0, text="fine thanks.", audio=clip)
    pf, pm = m.prompt_frames([seg], "and you?", speaker=1, voice_match=False, _codes={id(clip): codes})
    a, b = tok.encode("[0]fine thanks."), tok.encode("[1]and you?")
    assert pf.shape == pm.shape == (len(a) + 3 + 1 + len(b), n + 1)
    assert pf[: len(a), -1].tolist() == a and pm[: len(a), -1].all() and not pm[: len(a), :-1].any()
    aud = slice(len(a), len(a) + 4)
    np.testing.assert_array_equal(pf[aud, :-1][:3], codes.T)
    assert not pf[aud][3].any() and pm[aud, :-1].all() and not pm[aud, -1].any()  # the EOS frame: zeros WITH the audio mask set
    assert pf[len(a) + 4:, -1].tolist() == b and pm[len(a) + 4:, -1].all()
    # voice_match (the reference's default, sesame.py:727-744): ONE segment "ctx text + ' ' + text", the context audio, NO EOS frame
    vf, vm = m.prompt_frames([seg], "and you?", speaker=0, voice_match=True, _codes={id(clip): codes})
    j = tok.encode("[0]fine thanks. and you?")
    assert vf.shape == (len(j) + 3, n + 1) and vf[: len(j), -1].tolist() == j
    np.testing.assert_array_equal(vf[len(j):, :-1], codes.T)
    assert vm[len(j):, :-1].all() and not vm[len(j):, -1].any()
    # no tokenizer configured: a string is refused with the reason, ids still work
    m2 = Model(P.csm_tiny_config())
    with pytest.raises(ValueError, match="tokenizer"):
        m2._tokenize_text_segment("a string", 0)
    f2, _ = m2._tokenize_text_segment([5, 6, 7], 0)
    assert f2[:, -1].tolist() == [5, 6, 7]
