"""AED wrapper description (reference: trainer_3m_fix/model/conformer_aed_fmoe_localComm_catEmbed_domain_acc_hier.py:13-20):
only ``.encoder`` is built into the engine (builder.py:75); the attention decoders are outside the inference path, their
checkpoint entries are ignored."""
from collections import OrderedDict

from model.conformer_fmoe_localComm_catEmbed_domain_acc_hier import Net as ConformerEncoder


class Net:
    def __init__(self, input_dim, output_dim, encoder_conf=None, decoder_type="transformer", decoder_conf=None,
                 reverse_weight=0.0, padding_idx=None):
        self.input_dim, self.output_dim = input_dim, output_dim
        self.encoder = ConformerEncoder(input_dim, output_dim, **(encoder_conf or {}))

    def load_state_dict(self, state_dict, strict=True):
        enc = OrderedDict((k[len("encoder."):], v) for k, v in state_dict.items() if k.startswith("encoder."))
        return self.encoder.load_state_dict(enc if enc else state_dict, strict)

    def state_dict(self):
        return OrderedDict(("encoder." + k, v) for k, v in self.encoder.state_dict().items())

    def parameters(self):
        return self.encoder.parameters()

    def eval(self):
        return self
