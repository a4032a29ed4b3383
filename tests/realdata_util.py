"""Shared by tools/make_realdata.py (build container, runs the reference) and tests/test_realdata.py: the seed-regenerable
tiny encoders and the locally trained tokenizer of the real-data fixtures (tests/golden/realdata*.{npz,json})."""
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT = GOLDEN
VOCAB = 4096


def realdata_encoders():
    """The two tiny encoders of the fixtures, from a seed (the tests call this too)."""
    import torch
    from rankpo_amd import encoder as PE
    torch.manual_seed(4321)
    bert = PE.BertEncoder(PE.bert_config(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=VOCAB, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                         num_attention_heads=4, max_position_embeddings=256, pad_token_id=0))
    torch.manual_seed(4322)
    llama = PE.LlamaEncoder(PE.llama_config(vocab_size=VOCAB, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                            num_attention_heads=4, num_key_value_heads=2, pad_token_id=0,
                                            max_position_embeddings=1024, rope_theta=10000.0))
    return bert, llama


def weights_checksum(module):
    import torch
    with torch.no_grad():
        return float(sum((p.double() * torch.arange(1, p.numel() + 1, dtype=torch.float64).reshape(p.shape).remainder(7.0)
                          ).sum() for p in module.state_dict().values()))


def load_tokenizer(path=None):
    from transformers import PreTrainedTokenizerFast
    path = path or os.path.join(OUT, "realdata_tokenizer.json")
    return PreTrainedTokenizerFast(tokenizer_file=path, pad_token="<pad>", bos_token="<s>", eos_token="</s>")


