"""Text-format reader for the NetParameter subset the FlowNet2 deploy / train prototxts use, and the template substitution of the
reference's runner.

    scripts/run-flownet.py:38-58   $TARGET_WIDTH$ $TARGET_HEIGHT$ $ADAPTED_WIDTH$ $ADAPTED_HEIGHT$ $SCALE_WIDTH$ $SCALE_HEIGHT$
                                   are replaced line by line with str(value) before the file goes to caffe.Net
    src/caffe/util/upgrade_proto.cpp / io.cpp:ReadProtoFromTextFile -> google::protobuf::TextFormat::Parse of a caffe.proto
                                   NetParameter (caffe.proto:66-98: name, input, input_shape, input_dim, state, layer, layers)

The grammar handled is protobuf's text format as these files use it: `key: scalar`, `key { ... }`, `key: { ... }`, `key: [a, b]`,
`# comments`, strings in single or double quotes with the C escapes, numbers (decimal, hex, octal, floats incl. `inf` / `nan` /
exponent / trailing `f`), identifiers (enum names, true / false).  A message is a `Message`: an ordered multi-map (every field may
repeat).  No schema beyond which fields repeat in which message type (REPEATED_IN below, from caffe.proto); `to_dict` folds a
message into the plain dicts flownet2_amd.layers.LayerParameter carries.
"""
from __future__ import annotations

import math
import re
from typing import Any, Dict, Iterator, List, Tuple


class ParseError(ValueError):
    pass


class Message:
    """Ordered multi-map of one protobuf message."""

    __slots__ = ("fields",)

    def __init__(self):
        self.fields: List[Tuple[str, Any]] = []

    def add(self, key: str, value: Any):
        self.fields.append((key, value))

    def all(self, key: str) -> List[Any]:
        return [v for k, v in self.fields if k == key]

    def has(self, key: str) -> bool:
        return any(k == key for k, _ in self.fields)

    def get(self, key: str, default: Any = None) -> Any:
        """The LAST occurrence (protobuf semantics of a singular scalar field set twice)."""
        out = default
        for k, v in self.fields:
            if k == key:
                out = v
        return out

    def keys(self) -> List[str]:
        seen: List[str] = []
        for k, _ in self.fields:
            if k not in seen:
                seen.append(k)
        return seen

    def __repr__(self):
        return "Message(%s)" % ", ".join("%s=%r" % kv for kv in self.fields)


class Enum(str):
    """An identifier value (enum constant): a str that prints without quotes."""


_TOKEN = re.compile(r"""
    (?P<ws>\s+|\#[^\n]*)
  | (?P<str>"(?:[^"\\\n]|\\.)*"|'(?:[^'\\\n]|\\.)*')
  | (?P<num>[-+]?(?:0[xX][0-9a-fA-F]+|(?:\d+\.?\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?)[fF]?)|[-+](?:inf(?:inity)?|nan)\b)
  | (?P<id>[A-Za-z_][A-Za-z0-9_.]*)
  | (?P<punct>[{}<>:\[\],;])
""", re.VERBOSE)

_ESC = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", "'": "'", '"': '"', "a": "\a", "b": "\b", "f": "\f", "v": "\v", "?": "?"}


def _unescape(body: str) -> str:
    out, i = [], 0
    while i < len(body):
        c = body[i]
        if c != "\\":
            out.append(c)
            i += 1
            continue
        i += 1
        c = body[i]
        if c in _ESC:
            out.append(_ESC[c]); i += 1
        elif c in "xX":
            j = i + 1
            while j < len(body) and j < i + 3 and body[j] in "0123456789abcdefABCDEF":
                j += 1
            out.append(chr(int(body[i + 1:j], 16))); i = j
        elif c in "01234567":
            j = i
            while j < len(body) and j < i + 3 and body[j] in "01234567":
                j += 1
            out.append(chr(int(body[i:j], 8))); i = j
        else:
            raise ParseError("bad escape \\%s in string" % c)
    return "".join(out)


def _tokens(text: str) -> Iterator[Tuple[str, str, int]]:
    pos, line = 0, 1
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise ParseError("line %d: unexpected character %r" % (line, text[pos]))
        kind = m.lastgroup
        tok = m.group(kind)
        if kind != "ws":
            yield kind, tok, line
        line += tok.count("\n")
        pos = m.end()
    yield "eof", "", line


def _number(tok: str):
    t = tok.lower()
    neg = t.startswith("-")
    body = t.lstrip("+-")
    if body in ("inf", "infinity"):
        return -math.inf if neg else math.inf
    if body == "nan":
        return math.nan
    if body.startswith("0x"):
        return (-1 if neg else 1) * int(body, 16)
    if re.fullmatch(r"\d+", body):
        if len(body) > 1 and body[0] == "0":
            return (-1 if neg else 1) * int(body, 8)
        return int(t)
    return float(t.rstrip("f"))


class _Parser:
    def __init__(self, text: str):
        self.toks = list(_tokens(text))
        self.i = 0

    def peek(self):
        return self.toks[self.i]

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def scalar(self):
        kind, tok, line = self.next()
        if kind == "str":
            s = _unescape(tok[1:-1])
            while self.peek()[0] == "str":                 # adjacent string literals concatenate
                s += _unescape(self.next()[1][1:-1])
            return s
        if kind == "num":
            return _number(tok)
        if kind == "id":
            if tok in ("true", "True", "t"):
                return True
            if tok in ("false", "False", "f"):
                return False
            low = tok.lower()
            if low in ("inf", "infinity", "nan"):
                return _number(low)
            return Enum(tok)
        raise ParseError("line %d: expected a value, got %r" % (line, tok))

    def message(self, closer: str) -> Message:
        m = Message()
        while True:
            kind, tok, line = self.next()
            if kind == "eof":
                if closer:
                    raise ParseError("line %d: missing %r" % (line, closer))
                return m
            if kind == "punct" and tok == closer:
                return m
            if kind == "punct" and tok in ",;":
                continue
            if kind != "id":
                raise ParseError("line %d: expected a field name, got %r" % (line, tok))
            key = tok
            k2, t2, l2 = self.peek()
            colon = k2 == "punct" and t2 == ":"
            if colon:
                self.next()
                k2, t2, l2 = self.peek()
            if not colon and not (k2 == "punct" and t2 in "{<"):
                raise ParseError("line %d: expected ':' or '{' after field name %r" % (line, key))      # only messages may omit the colon
            if k2 == "punct" and t2 in "{<":
                self.next()
                m.add(key, self.message("}" if t2 == "{" else ">"))
            elif k2 == "punct" and t2 == "[":
                self.next()
                while True:
                    k3, t3, l3 = self.peek()
                    if k3 == "punct" and t3 == "]":
                        self.next()
                        break
                    if k3 == "punct" and t3 == ",":
                        self.next()
                        continue
                    if k3 == "punct" and t3 in "{<":
                        self.next()
                        m.add(key, self.message("}" if t3 == "{" else ">"))
                    else:
                        m.add(key, self.scalar())
            else:
                m.add(key, self.scalar())


def parse(text: str) -> Message:
    """Text-format message -> Message.  Raises ParseError with a line number."""
    return _Parser(text).message("")


# ---- template variables (scripts/run-flownet.py:38-58) ---------------------------------------------------------------
def deploy_vars(width: int, height: int, divisor: float = 64.0) -> Dict[str, Any]:
    aw, ah = int(math.ceil(width / divisor) * divisor), int(math.ceil(height / divisor) * divisor)
    return {"TARGET_WIDTH": width, "TARGET_HEIGHT": height, "ADAPTED_WIDTH": aw, "ADAPTED_HEIGHT": ah,
            "SCALE_WIDTH": width / float(aw), "SCALE_HEIGHT": height / float(ah)}


def substitute(template: str, variables: Dict[str, Any]) -> str:
    """`$KEY$` -> str(value), line by line like the reference (python's str(float): the shortest repr that round-trips)."""
    out = []
    for line in template.splitlines(keepends=True):
        for key, value in variables.items():
            line = line.replace("$%s$" % key, str(value))
        out.append(line)
    return "".join(out)


def unresolved(text: str) -> List[str]:
    return sorted(set(re.findall(r"\$([A-Z_]+)\$", text)))


# ---- Message -> plain dicts ------------------------------------------------------------------------------------------
# The repeated fields of caffe.proto, per MESSAGE TYPE: the key is the field name through which the message is reached ("" = the root
# NetParameter).  A field name alone does not decide it: `mean` is `repeated float` in AugmentationParameter (caffe.proto:498) and an
# `optional float` in RandomGeneratorParameter (caffe.proto:610: `translate { rand_type: "uniform" mean: 0 spread: 0.4 }`), `param` is a
# ParamSpec list in LayerParameter and a string list in V1LayerParameter, `shape` repeats in InputParameter and not in ReshapeParameter.
_LAYER_REPEATED = {"bottom", "top", "loss_weight", "param", "blobs", "include", "exclude", "propagate_down"}
REPEATED_IN = {
    "": {"input", "input_shape", "input_dim", "layer", "layers"},                                      # NetParameter, caffe.proto:66-98
    "layer": _LAYER_REPEATED,                                                                          # LayerParameter, :315-349
    "layers": _LAYER_REPEATED | {"blobs_lr", "weight_decay", "blob_share_mode"},                       # V1LayerParameter, :1533-1590
    "input_shape": {"dim"}, "shape": {"dim"},                                                          # BlobShape, :7
    "blobs": {"data", "diff", "double_data", "double_diff"},                                           # BlobProto, :12-15
    "state": {"stage"}, "include": {"stage", "not_stage"}, "exclude": {"stage", "not_stage"},          # NetState / NetStateRule, :262-280
    "augmentation_param": {"mean", "chromatic_eigvec"},                                                # AugmentationParameter, :498-504
    "weight_filler": {"diag_val"}, "bias_filler": {"diag_val"}, "data_filler": {"diag_val"},           # FillerParameter, :63
    "convolution_param": {"kernel_size", "pad", "stride", "dilation"},                                 # ConvolutionParameter, :853-859
    "eltwise_param": {"coeff"},                                                                        # EltwiseParameter, :1018
    "slice_param": {"slice_point"},                                                                    # SliceParameter, :1438
    "data_param": {"slice_point", "encoding", "subtract"},                                             # DataParameter, :979-983
    "dummy_data_param": {"data_filler", "shape", "num", "channels", "height", "width"},                # DummyDataParameter, :1001-1008
    "input_param": {"shape"},                                                                          # InputParameter, :1155
    "mean_param": {"value"},                                                                           # MeanParameter, :687
    "transform_param": {"mean_value"},                                                                 # TransformationParameter, :715
    "crop_param": {"offset"},                                                                          # CropParameter, :915
    "lpq_loss_param": {"pq_episode_starts_at_iter", "p", "q"},                                         # LpqLossParameter, :599-601
}
# every name that is repeated somewhere (kept for callers that only ask "can this field repeat at all")
REPEATED = set().union(*REPEATED_IN.values())


def to_dict(m: Message, parent: str = "") -> Dict[str, Any]:
    """Nested plain dict: sub-messages become dicts; the repeated fields of the message type reached through `parent` become lists
    (also with one or no bracket), and so does any other field that occurs more than once."""
    out: Dict[str, Any] = {}
    rep = REPEATED_IN.get(parent, ())
    for key in m.keys():
        vals = [to_dict(v, key) if isinstance(v, Message) else v for v in m.all(key)]
        out[key] = vals if (key in rep or len(vals) > 1) else vals[0]
    return out
