// onnx_verify.cpp — is the graph in this `.onnx` the computation the fixed launch schedule (voice.hip) performs?
//
// The reference EXECUTES whatever the graph says, node by node (GraphExecutor.swift:227-265, arms :591-2662). This library does not:
// it takes the initializers and runs its own schedule, so an export whose op order, activation slope, LayerNorm epsilon, padding,
// mask handling or an extra node differs would render wrong audio without any error. This verifier closes that hole (SURVEY.md §8f row
// 1's pattern matcher in its minimal form; VERDICT r2 missing #1): it walks the NodeProtos by DATAFLOW (producer / consumer edges, not
// node positions or names, so the exporter's shape plumbing — Shape / Gather / Concat / Cast / Unsqueeze … — and mask multiplies
// are looked through) and checks, block by block, what the schedule assumes. The first difference is reported with the node it was
// found at and the load is refused (PIPER_HIP_ERR_UNSUPPORTED). Host-only code; no GPU.
//
// What is checked (each item = an assumption of voice.hip):
//   header   opset 15, inputs input / input_lengths / scales, output `output`, first node Gather (ONNXParsingTests.swift:22-36)
//   census   every op type is one of the reference's 50 arms (GraphExecutor.swift:592-2659, default → unsupportedOp :2661)
//   convs    each voice Conv / ConvTranspose: op type, strides, dilations, group, auto_pad, EFFECTIVE padding (node pads + an explicit
//            Pad in front, as VITS' FFN uses) = the `same` padding the kernels apply
//   encoder  Gather·√H; per layer q/k/v → QKᵀ with q/√d, rel-K logits through the Pad/Reshape/Pad/Reshape/Slice skew, Softmax on the
//            last axis, P·V + rel-V through the inverse skew, conv_o, residual Add, LayerNorm chain with ε = 1e-5, FFN conv_1 → Relu → conv_2
//   flow     n_flows Flips (step −1 Slice on the channel axis) each feeding a coupling's Split; per WaveNet layer in_layer → tanh(first
//            half)·sigmoid(second half) → res_skip; x1 − m with m from `post`
//   decoder  LeakyRelu(0.1) before every ConvTranspose and ResBlock conv, residual Adds chained through the ResBlock, the MRF mean
//            as a division by the ResBlock count, LeakyRelu(0.01) → conv_post → Tanh → `output`
//   noise    one RandomNormalLike for z (+ one for the duration predictor)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <set>
#include <unordered_map>

#include "../../include/piper_hip_voice_layout.h"
#include "onnx_model.h"

using namespace ph;
using namespace ph::onnx;

namespace {

const char* const kSupportedOps[] = {
    "Gather", "GatherElements", "Mul", "Div", "Sub", "Add", "Transpose", "Shape", "Range", "Unsqueeze", "Concat", "Reshape", "Pad", "Clip", "Slice",
    "Less", "GreaterOrEqual", "LessOrEqual", "And", "Not", "Cast", "Equal", "Where", "Conv", "ConvTranspose", "MatMul", "Softmax", "Relu", "Erf",
    "Softplus", "Neg", "Exp", "Ceil", "Tanh", "Sigmoid", "LeakyRelu", "Pow", "Sqrt", "ReduceMean", "ReduceSum", "ReduceMax", "Split",
    "ConstantOfShape", "Expand", "ScatterND", "Squeeze", "NonZero", "GatherND", "CumSum", "RandomNormalLike",
    "Constant", "Identity"};  // the last two hold no arithmetic (folded into initializers / aliases by every ONNX runtime)

struct Verifier {
  const piper_hip_onnx* m;
  const piper_hip_voice_config* c;
  std::unordered_map<std::string, int> prod;
  std::unordered_map<std::string, std::vector<int>> cons;
  std::unordered_map<std::string, int> conv_by_w, node_by_name;
  std::string why;

  const Node& N(int i) const { return m->nodes[(size_t)i]; }
  std::string label(int i) const {
    if (i < 0) return "(graph input or initializer)";
    const Node& n = N(i);
    return (n.name.empty() ? "#" + std::to_string(i) : "'" + n.name + "'") + " (" + n.op + ")";
  }
  bool fail(const std::string& s) {
    if (why.empty()) why = s;
    return false;
  }
  int producer(const std::string& t) const {
    auto it = prod.find(t);
    return it == prod.end() ? -1 : it->second;
  }
  bool is_init(const std::string& t) const { return m->by_name.count(t) != 0; }

  // numbers of a small constant: an initializer or a Constant node's `value`
  bool const_values(const std::string& t, std::vector<double>& out) const {
    const Tensor* ten = nullptr;
    auto it = m->by_name.find(t);
    if (it != m->by_name.end()) ten = &m->tensors[(size_t)it->second];
    else {
      const int p = producer(t);
      if (p >= 0 && N(p).op == "Constant")
        if (const Attr* a = N(p).attr("value"))
          if (a->has_t) ten = &a->t;
      if (p >= 0 && N(p).op == "Identity" && !N(p).inputs.empty()) return const_values(N(p).inputs[0], out);
    }
    if (!ten) return false;
    const int64_t n = ten->count();
    if (n < 0 || n > 64) return false;
    out.clear();
    if (ten->dtype == 1) {
      const uint8_t* src = ten->raw_len ? ten->raw : ten->fdat;
      const size_t len = ten->raw_len ? ten->raw_len : ten->fdat_len;
      if (src && len == (size_t)n * 4) {
        for (int64_t i = 0; i < n; i++) { float f; memcpy(&f, src + 4 * i, 4); out.push_back(f); }
      } else if ((int64_t)ten->floats_unpacked.size() == n) {
        for (float f : ten->floats_unpacked) out.push_back(f);
      } else return false;
    } else if (ten->dtype == 7) {
      if (ten->raw_len == (size_t)n * 8) {
        for (int64_t i = 0; i < n; i++) { int64_t v; memcpy(&v, ten->raw + 8 * i, 8); out.push_back((double)v); }
      } else if ((int64_t)ten->i64_unpacked.size() == n) {
        for (int64_t v : ten->i64_unpacked) out.push_back((double)v);
      } else return false;
    } else return false;
    return true;
  }
  bool scalar(const std::string& t, double& v) const {
    std::vector<double> xs;
    if (!const_values(t, xs) || xs.size() != 1) return false;
    v = xs[0];
    return true;
  }
  static bool close(double a, double b, double rel = 1e-3) { return std::fabs(a - b) <= rel * std::max(1e-30, std::fabs(b)); }

  // a tensor derived from a sequence mask (comparison → Cast / Unsqueeze / products of masks)
  bool mask_like(const std::string& t, int depth = 0) const {
    const int p = producer(t);
    if (p < 0 || depth > 8) return false;
    const Node& n = N(p);
    static const std::set<std::string> cmp = {"Less", "LessOrEqual", "GreaterOrEqual", "Equal", "Not", "And"};
    static const std::set<std::string> thru = {"Cast", "Unsqueeze", "Squeeze", "Identity", "Expand", "Slice", "Reshape"};
    if (cmp.count(n.op)) return true;
    if (thru.count(n.op)) return !n.inputs.empty() && mask_like(n.inputs[0], depth + 1);
    if (n.op == "Mul") return n.inputs.size() == 2 && mask_like(n.inputs[0], depth + 1) && mask_like(n.inputs[1], depth + 1);
    return false;
  }
  static bool in(const std::string& op, std::initializer_list<const char*> ops) {
    for (const char* o : ops)
      if (op == o) return true;
    return false;
  }
  // Walk from tensor `t` towards its producer through nodes that do not change values on the valid positions: aliases, casts, mask
  // multiplies, and — accumulating what they pad on the last axis — explicit Pads. `extra` = further op types to look through.
  // Returns the first other node (−1: a graph input / initializer); `at` = the tensor that node produced.
  int back(std::string t, std::initializer_list<const char*> extra = {}, int64_t* pad_l = nullptr, int64_t* pad_r = nullptr, std::string* at = nullptr) const {
    for (int hop = 0; hop < 16; hop++) {
      const int p = producer(t);
      if (at) *at = t;
      if (p < 0) return -1;
      const Node& n = N(p);
      if (n.inputs.empty()) return p;
      if (in(n.op, {"Identity", "Cast", "Unsqueeze", "Squeeze"}) || in(n.op, extra)) { t = n.inputs[0]; continue; }
      if (n.op == "Mul" && n.inputs.size() == 2) {
        if (mask_like(n.inputs[1])) { t = n.inputs[0]; continue; }
        if (mask_like(n.inputs[0])) { t = n.inputs[1]; continue; }
        return p;
      }
      if (n.op == "Pad" && (pad_l || pad_r)) {
        std::vector<double> pads;
        if (n.inputs.size() >= 2 && const_values(n.inputs[1], pads) && pads.size() >= 2 && pads.size() % 2 == 0) {
          const size_t r = pads.size() / 2;
          bool only_last = true;
          for (size_t i = 0; i < pads.size(); i++)
            if (i != r - 1 && i != 2 * r - 1 && pads[i] != 0) only_last = false;
          if (only_last) {
            if (pad_l) *pad_l += (int64_t)pads[r - 1];
            if (pad_r) *pad_r += (int64_t)pads[2 * r - 1];
            t = n.inputs[0];
            continue;
          }
        }
        return p;
      }
      return p;
    }
    return -1;
  }
  // consumers of `t`, looking through the same value-preserving nodes
  void fwd(const std::string& t, std::vector<int>& out, std::initializer_list<const char*> extra = {}, int depth = 0) const {
    auto it = cons.find(t);
    if (it == cons.end() || depth > 12) return;
    for (int ci : it->second) {
      const Node& n = N(ci);
      bool thru = in(n.op, {"Identity", "Cast", "Unsqueeze", "Squeeze"}) || in(n.op, extra);
      if (n.op == "Mul" && n.inputs.size() == 2) {
        const std::string& other = n.inputs[0] == t ? n.inputs[1] : n.inputs[0];
        if (mask_like(other)) thru = true;
      }
      if (thru && !n.outputs.empty()) fwd(n.outputs[0], out, extra, depth + 1);
      else out.push_back(ci);
    }
  }
  int conv_of(const std::string& module) const {  // the node that consumes a module's weight (by initializer name, weight_v, or scope name)
    for (const char* suf : {".weight", ".weight_v"}) {
      auto it = conv_by_w.find(module + suf);
      if (it != conv_by_w.end()) return it->second;
    }
    std::string sc;
    size_t i = 0;
    while (i < module.size()) {
      size_t j = module.find('.', i);
      if (j == std::string::npos) j = module.size();
      const std::string part = module.substr(i, j - i);
      const bool numeric = !part.empty() && part.find_first_not_of("0123456789") == std::string::npos;
      sc += (numeric && !sc.empty()) ? "." : "/";
      sc += part;
      i = j + 1;
    }
    for (const char* suf : {"/Conv", "/ConvTranspose"}) {
      auto it = node_by_name.find(sc + suf);
      if (it != node_by_name.end()) return it->second;
    }
    return -1;
  }
  static int64_t attr_i(const Node& n, const char* name, int64_t dflt) {
    const Attr* a = n.attr(name);
    if (!a) return dflt;
    if (a->has_i) return a->i;
    return a->ints.empty() ? dflt : a->ints[0];
  }

  // ---- one Conv / ConvTranspose against the geometry the schedule applies ----
  bool check_conv(const std::string& module, bool transpose, int64_t k, int64_t stride, int64_t dil, int64_t pl, int64_t pr, int* node_out = nullptr) {
    const int ni = conv_of(module);
    if (ni < 0) return fail("no Conv node consumes '" + module + ".weight'");
    if (node_out) *node_out = ni;
    const Node& n = N(ni);
    if (n.op != (transpose ? "ConvTranspose" : "Conv")) return fail(label(ni) + ": expected " + (transpose ? "ConvTranspose" : "Conv") + " for " + module);
    if (const Attr* ap = n.attr("auto_pad"))
      if (!ap->s.empty() && ap->s != "NOTSET") return fail(label(ni) + ": auto_pad=" + ap->s + " (the reference ignores auto_pad, GraphExecutor.swift:1746-1753; only explicit pads are supported)");
    if (attr_i(n, "group", 1) != 1) return fail(label(ni) + ": group " + std::to_string(attr_i(n, "group", 1)) + ", expected 1");
    if (attr_i(n, "strides", 1) != stride) return fail(label(ni) + ": stride " + std::to_string(attr_i(n, "strides", 1)) + ", the schedule applies " + std::to_string(stride));
    if (attr_i(n, "dilations", 1) != dil) return fail(label(ni) + ": dilation " + std::to_string(attr_i(n, "dilations", 1)) + ", the schedule applies " + std::to_string(dil));
    if (const Attr* ks = n.attr("kernel_shape"))
      if (!ks->ints.empty() && ks->ints[0] != k) return fail(label(ni) + ": kernel_shape " + std::to_string(ks->ints[0]) + ", the weight has " + std::to_string(k) + " taps");
    if (transpose && attr_i(n, "output_padding", 0) != 0) return fail(label(ni) + ": output_padding is not 0");
    int64_t npl = 0, npr = 0;
    if (const Attr* pa = n.attr("pads"))
      if (pa->ints.size() >= 2) { npl = pa->ints[0]; npr = pa->ints[pa->ints.size() / 2]; }
    if (!transpose) (void)back(n.inputs[0], {}, &npl, &npr);  // + an explicit Pad in front (VITS FFN: F.pad, then a conv without padding)
    if (npl != pl || npr != pr)
      return fail(label(ni) + ": effective padding (" + std::to_string(npl) + ", " + std::to_string(npr) + "), the schedule applies (" + std::to_string(pl) + ", " +
                  std::to_string(pr) + ")");
    return true;
  }

  // `t` must be LeakyRelu(alpha) of something: returns that node or −1 (failure recorded)
  int expect_lrelu(const std::string& t, double alpha, const std::string& what) {
    const int a = back(t);
    if (a < 0 || N(a).op != "LeakyRelu") { fail(what + ": its input comes from " + label(a) + ", expected LeakyRelu(" + std::to_string(alpha) + ")"); return -1; }
    const Attr* at = N(a).attr("alpha");
    const double got = at && at->has_f ? at->f : 0.01;  // ONNX default (GraphExecutor.swift:2047-2069)
    if (!close(got, alpha, 1e-4)) { fail(label(a) + ": alpha " + std::to_string(got) + ", the schedule applies " + std::to_string(alpha) + " (" + what + ")"); return -1; }
    return a;
  }

  bool run() {
    const size_t nn = m->nodes.size();
    for (size_t i = 0; i < nn; i++) {
      const Node& n = m->nodes[i];
      for (const std::string& o : n.outputs)
        if (!o.empty()) prod[o] = (int)i;
      for (const std::string& in_ : n.inputs)
        if (!in_.empty()) cons[in_].push_back((int)i);
      if ((n.op == "Conv" || n.op == "ConvTranspose") && n.inputs.size() >= 2) conv_by_w[n.inputs[1]] = (int)i;
      if (!n.name.empty()) node_by_name[n.name] = (int)i;
    }
    // ---- arity: after this pass every node has an output, every non-Constant node an input, every binary op two (the walks below
    // index inputs[0] / inputs[1] / outputs[0] without further checks; hostile files are fuzzed through here under ASAN) ----
    for (size_t i = 0; i < nn; i++) {
      const Node& n = m->nodes[i];
      const size_t need = n.op == "Constant" ? 0 : n.op == "Where" ? 3 : in(n.op, {"Conv", "ConvTranspose", "MatMul", "Add", "Sub", "Mul", "Div", "Pow", "Gather", "Equal", "Less"}) ? 2 : 1;
      if (n.outputs.empty() || n.outputs[0].empty() || n.inputs.size() < need) return fail(label((int)i) + ": malformed node (" + std::to_string(n.inputs.size()) + " inputs, " + std::to_string(n.outputs.size()) + " outputs)");
    }
    // ---- header (ONNXParsingTests.swift:22-36) ----
    if (m->opset != 15) return fail("opset " + std::to_string(m->opset) + ": Piper exports (and the reference's op semantics: axes-as-input Unsqueeze / Split, GraphExecutor.swift:982-988) are opset 15");
    std::vector<std::string> gin;
    for (const std::string& s : m->graph_inputs)
      if (!is_init(s)) gin.push_back(s);  // old exporters list initializers as inputs too
    if (gin != std::vector<std::string>{"input", "input_lengths", "scales"}) {
      std::string got;
      for (const std::string& s : gin) got += (got.empty() ? "" : ", ") + s;
      return fail("graph inputs are [" + got + "], expected [input, input_lengths, scales]" + (std::count(gin.begin(), gin.end(), "sid") ? " (multi-speaker voices are not supported)" : ""));
    }
    if (m->graph_outputs != std::vector<std::string>{"output"}) return fail("graph outputs are not [output]");
    if (nn == 0 || m->nodes[0].op != "Gather") return fail("first node is " + (nn ? label(0) : std::string("missing")) + ", expected the embedding Gather");
    // ---- census ----
    {
      std::set<std::string> ok(std::begin(kSupportedOps), std::end(kSupportedOps));
      for (size_t i = 0; i < nn; i++)
        if (!ok.count(m->nodes[i].op)) return fail(label((int)i) + ": op type outside the reference's op set (GraphExecutor.swift:2661 unsupportedOp)");
    }
    const int H = c->hidden, d = c->hidden / c->n_heads;
    char p[128];
    // ---- embedding ----
    {
      const Node& g0 = m->nodes[0];
      if (g0.inputs.size() < 2 || g0.inputs[0] != "enc_p.emb.weight" || g0.inputs[1] != "input") return fail(label(0) + ": expected Gather(enc_p.emb.weight, input)");
      std::vector<int> cs;
      fwd(g0.outputs[0], cs);
      double s = 0;
      bool ok = false;
      for (int ci : cs)
        if (N(ci).op == "Mul" && N(ci).inputs.size() == 2 && (scalar(N(ci).inputs[1], s) || scalar(N(ci).inputs[0], s)) && close(s, std::sqrt((double)H))) ok = true;
      if (!ok) return fail(label(0) + ": the embedding is not scaled by sqrt(hidden) = " + std::to_string(std::sqrt((double)H)));
    }
    // ---- text encoder layers ----
    for (int l = 0; l < c->n_layers; l++) {
      int nq, nk, nv, no, n1, n2;
      snprintf(p, sizeof p, "enc_p.encoder.attn_layers.%d", l);
      const std::string A = p;
      if (!check_conv(A + ".conv_q", false, 1, 1, 1, 0, 0, &nq) || !check_conv(A + ".conv_k", false, 1, 1, 1, 0, 0, &nk) || !check_conv(A + ".conv_v", false, 1, 1, 1, 0, 0, &nv) ||
          !check_conv(A + ".conv_o", false, 1, 1, 1, 0, 0, &no))
        return false;
      snprintf(p, sizeof p, "enc_p.encoder.ffn_layers.%d", l);
      const std::string F = p;
      const int kf = c->ffn_kernel;
      if (!check_conv(F + ".conv_1", false, kf, 1, 1, (kf - 1) / 2, kf / 2, &n1) || !check_conv(F + ".conv_2", false, kf, 1, 1, (kf - 1) / 2, kf / 2, &n2)) return false;
      // conv_o ← (Transpose / Reshape) ← Add(P·V, rel-V term)
      const int add = back(N(no).inputs[0], {"Transpose", "Reshape"});
      if (add < 0 || N(add).op != "Add" || N(add).inputs.size() != 2) return fail(label(no) + ": its input comes from " + label(add) + ", expected the Add of P·V and the relative-value term");
      int softmax = -1, av = -1, relv = -1;
      for (int side = 0; side < 2; side++) {
        const int mm = back(N(add).inputs[(size_t)side]);
        if (mm < 0 || N(mm).op != "MatMul") return fail(label(add) + ": operand " + std::to_string(side) + " comes from " + label(mm) + ", expected MatMul");
        const int rhs = back(N(mm).inputs[1], {"Transpose", "Reshape"});
        if (rhs == nv) av = mm; else relv = mm;
      }
      if (av < 0 || relv < 0) return fail(label(add) + ": expected one MatMul with conv_v's output (P·V) and one with the relative-value embeddings");
      softmax = back(N(av).inputs[0]);
      if (softmax < 0 || N(softmax).op != "Softmax") return fail(label(av) + ": its left operand comes from " + label(softmax) + ", expected Softmax");
      {
        const int64_t ax = attr_i(N(softmax), "axis", -1);
        if (ax != -1 && ax != 3) return fail(label(softmax) + ": axis " + std::to_string(ax) + ", the attention softmax runs over the last axis (GraphExecutor.swift:1917-1929 accepts only that)");
      }
      {  // rel-V: MatMul(unskew(P), emb_rel_v window)
        std::string at;
        const int src = back(N(relv).inputs[1], {"Transpose", "Reshape", "Slice", "Pad"}, nullptr, nullptr, &at);
        if (src >= 0 || at != A + ".emb_rel_v") return fail(label(relv) + ": right operand does not come from " + A + ".emb_rel_v");
        int pads = 0, slices = 0, reshapes = 0;
        std::string t = N(relv).inputs[0];
        for (int hop = 0; hop < 12; hop++) {
          const int q = producer(t);
          if (q < 0) break;
          if (q == softmax) break;
          const std::string& op = N(q).op;
          if (op == "Pad") pads++; else if (op == "Slice") slices++; else if (op == "Reshape") reshapes++;
          else if (!in(op, {"Identity", "Cast"})) return fail(label(q) + ": unexpected node in the absolute→relative skew chain of layer " + std::to_string(l));
          t = N(q).inputs[0];
        }
        if (producer(t) != softmax || pads != 2 || reshapes != 2 || slices < 1)
          return fail(label(relv) + ": left operand is not the Pad/Reshape/Pad/Reshape/Slice skew of the softmax output (pads " + std::to_string(pads) + ", reshapes " + std::to_string(reshapes) + ", slices " + std::to_string(slices) + ")");
      }
      // Softmax ← [Where(mask, −1e4, ·)] ← Add(QKᵀ, skew(rel-K logits))
      int sadd = producer(N(softmax).inputs[0]);
      if (sadd >= 0 && N(sadd).op == "Where" && N(sadd).inputs.size() == 3) sadd = back(N(sadd).inputs[2]);
      if (sadd < 0 || N(sadd).op != "Add" || N(sadd).inputs.size() != 2) return fail(label(softmax) + ": its input comes from " + label(sadd) + ", expected Add(scores, relative logits) [behind the mask fill]");
      int qk = -1, relk = -1;
      for (int side = 0; side < 2; side++) {
        std::string t = N(sadd).inputs[(size_t)side];
        int pads = 0, reshapes = 0, slices = 0, q = -1;
        for (int hop = 0; hop < 12; hop++) {
          q = producer(t);
          if (q < 0) break;
          const std::string& op = N(q).op;
          if (op == "Pad") pads++; else if (op == "Slice") slices++; else if (op == "Reshape") reshapes++;
          else if (!in(op, {"Identity", "Cast"})) break;
          t = N(q).inputs[0];
        }
        if (q < 0 || N(q).op != "MatMul") return fail(label(sadd) + ": operand " + std::to_string(side) + " comes from " + label(q) + ", expected MatMul");
        if (pads == 0 && reshapes == 0 && slices == 0) qk = q;
        else if (pads == 2 && reshapes == 2 && slices >= 1) relk = q;
        else return fail(label(q) + ": reaches the score Add through " + std::to_string(pads) + " Pad / " + std::to_string(reshapes) + " Reshape / " + std::to_string(slices) +
                         " Slice nodes, expected the Pad/Reshape/Pad/Reshape/Slice relative→absolute skew");
      }
      if (qk < 0 || relk < 0) return fail(label(sadd) + ": expected QK^T plus the skewed relative-key logits");
      {
        const int kt = back(N(qk).inputs[1], {"Transpose", "Reshape"});
        if (kt != nk) return fail(label(qk) + ": right operand does not come from conv_k of layer " + std::to_string(l));
        std::string at;
        const int rk = back(N(relk).inputs[1], {"Transpose", "Reshape", "Slice", "Pad"}, nullptr, nullptr, &at);
        if (rk >= 0 || at != A + ".emb_rel_k") return fail(label(relk) + ": right operand does not come from " + A + ".emb_rel_k");
        for (int mmi : {qk, relk}) {  // the query enters both products divided by sqrt(head_dim)
          const int sc = back(N(mmi).inputs[0]);
          double s = 0;
          bool ok = false;
          if (sc >= 0 && N(sc).inputs.size() == 2 && scalar(N(sc).inputs[1], s)) {
            if (N(sc).op == "Div") ok = close(s, std::sqrt((double)d));
            if (N(sc).op == "Mul") ok = close(s, 1.0 / std::sqrt((double)d));
          }
          if (!ok) return fail(label(mmi) + ": the query is not scaled by 1/sqrt(head_dim = " + std::to_string(d) + ") (found " + label(sc) + ", constant " + std::to_string(s) + ")");
          if (back(N(sc).inputs[0], {"Transpose", "Reshape"}) != nq) return fail(label(sc) + ": does not scale conv_q's output of layer " + std::to_string(l));
        }
      }
      // FFN: conv_1 → Relu → conv_2
      {
        const int act = back(N(n2).inputs[0], {"Pad"});
        if (act < 0 || N(act).op != "Relu") return fail(label(n2) + ": its input comes from " + label(act) + ", expected Relu(conv_1) (GraphExecutor.swift:1931-1944)");
        if (back(N(act).inputs[0]) != n1) return fail(label(act) + ": does not take conv_1 of layer " + std::to_string(l));
      }
      // the two LayerNorms behind conv_o's and conv_2's residual Adds: epsilon of the ReduceMean … Sqrt chain
      for (int src : {no, n2}) {
        std::vector<int> cs;
        fwd(N(src).outputs[0], cs);
        int radd = -1;
        for (int ci : cs)
          if (N(ci).op == "Add") radd = ci;
        if (radd < 0) return fail(label(src) + ": its output is not added to the residual stream");
        // breadth-first, a few nodes deep: the Sqrt of this LayerNorm
        std::vector<int> frontier{radd};
        int sq = -1;
        for (int depth = 0; depth < 9 && sq < 0; depth++) {
          std::vector<int> next;
          for (int ni : frontier)
            for (const std::string& o : N(ni).outputs) {
              auto it = cons.find(o);
              if (it == cons.end()) continue;
              for (int ci : it->second) {
                if (N(ci).op == "Sqrt") sq = ci;
                next.push_back(ci);
              }
            }
          frontier.swap(next);
        }
        if (sq < 0) return fail(label(radd) + ": no LayerNorm (ReduceMean … Sqrt chain, GraphExecutor.swift:2071-2125) follows the residual Add");
        const int ea = producer(N(sq).inputs[0]);
        double eps = 0;
        if (ea < 0 || N(ea).op != "Add" || N(ea).inputs.size() != 2 || !(scalar(N(ea).inputs[1], eps) || scalar(N(ea).inputs[0], eps)))
          return fail(label(sq) + ": its input is not Add(variance, epsilon constant)");
        if (!close(eps, 1e-5, 1e-3)) return fail(label(ea) + ": LayerNorm epsilon " + std::to_string(eps) + ", the schedule applies 1e-5");
        const int rm = producer(N(ea).inputs[0]);
        if (rm < 0 || N(rm).op != "ReduceMean") return fail(label(ea) + ": variance does not come from ReduceMean");
        std::vector<double> axes;
        const Attr* ax = N(rm).attr("axes");
        const int64_t axv = ax && !ax->ints.empty() ? ax->ints[0] : -1;
        if (axv != -1 && axv != 2) return fail(label(rm) + ": ReduceMean over axis " + std::to_string(axv) + ", expected the last axis (the channel axis after the Transpose)");
      }
    }
    if (!check_conv("enc_p.proj", false, 1, 1, 1, 0, 0)) return false;
    // ---- noise ----
    {
      int rn = 0, first_extra = -1;
      for (size_t i = 0; i < nn; i++)
        if (m->nodes[i].op == "RandomNormalLike" && ++rn > 1 + (c->dp_present ? 1 : 0) && first_extra < 0) first_extra = (int)i;
      if (first_extra >= 0) return fail(label(first_extra) + ": more RandomNormalLike nodes than the schedule draws (z" + (c->dp_present ? " and the duration predictor" : "") + ")");
      if (rn < 1) return fail("no RandomNormalLike node: the prior sample z_p = m_p + noise * exp(logs_p) * noise_scale is missing");
    }
    // ---- flow (reverse) ----
    {
      int flips = 0;
      for (size_t i = 0; i < nn; i++) {
        const Node& n = m->nodes[i];
        double st = 0, ax = 0;
        if (n.op == "Slice" && n.inputs.size() >= 5 && scalar(n.inputs[4], st) && st == -1 && scalar(n.inputs[3], ax) && ax == 1) flips++;
      }
      if (flips != c->n_flows) return fail("the graph holds " + std::to_string(flips) + " channel Flips (Slice with step -1 on axis 1), the schedule applies " + std::to_string(c->n_flows));
    }
    for (int f = 0; f < c->n_flows; f++) {
      snprintf(p, sizeof p, "flow.flows.%d", 2 * f);
      const std::string Fl = p;
      int npre, npost;
      if (!check_conv(Fl + ".pre", false, 1, 1, 1, 0, 0, &npre) || !check_conv(Fl + ".post", false, 1, 1, 1, 0, 0, &npost)) return false;
      {  // pre ← Split ← Flip
        const int sp = back(N(npre).inputs[0]);
        if (sp < 0 || !in(N(sp).op, {"Split", "Slice"})) return fail(label(npre) + ": its input comes from " + label(sp) + ", expected the first half of the coupling's Split");
        const int fl = back(N(sp).inputs[0]);
        double st = 0;
        if (fl < 0 || N(fl).op != "Slice" || N(fl).inputs.size() < 5 || !scalar(N(fl).inputs[4], st) || st != -1)
          return fail(label(sp) + ": the coupling does not read a Flip (found " + label(fl) + ")");
      }
      {  // x1 − m
        std::vector<int> cs;
        fwd(N(npost).outputs[0], cs);
        bool ok = false;
        for (int ci : cs)
          if (N(ci).op == "Sub" && N(ci).inputs.size() == 2 && back(N(ci).inputs[1]) == npost) ok = true;
        if (!ok) return fail(label(npost) + ": its output is not the subtrahend of the coupling's x1 - m");
      }
      const int kw = c->wn_kernel;
      for (int i = 0; i < c->wn_layers; i++) {
        int nin, nrs;
        snprintf(p, sizeof p, "%s.enc.in_layers.%d", Fl.c_str(), i);
        if (!check_conv(p, false, kw, 1, 1, (kw - 1) / 2, (kw - 1) / 2, &nin)) return false;
        snprintf(p, sizeof p, "%s.enc.res_skip_layers.%d", Fl.c_str(), i);
        if (!check_conv(p, false, 1, 1, 1, 0, 0, &nrs)) return false;
        // res_skip ← Mul(Tanh(first half), Sigmoid(second half)) of in_layer (+ the zero conditioning Add)
        const int gate = back(N(nrs).inputs[0]);
        if (gate < 0 || N(gate).op != "Mul" || N(gate).inputs.size() != 2) return fail(label(nrs) + ": its input comes from " + label(gate) + ", expected the gate Mul(tanh, sigmoid)");
        int th = -1, sg = -1;
        for (int side = 0; side < 2; side++) {
          const int a = producer(N(gate).inputs[(size_t)side]);
          if (a >= 0 && N(a).op == "Tanh") th = a;
          if (a >= 0 && N(a).op == "Sigmoid") sg = a;
        }
        if (th < 0 || sg < 0) return fail(label(gate) + ": expected Tanh x Sigmoid (GraphExecutor.swift:2017-2045)");
        for (int a : {th, sg}) {
          const int sl = back(N(a).inputs[0]);
          double start = -1;
          bool ok = false;
          if (sl >= 0 && N(sl).op == "Slice" && N(sl).inputs.size() >= 2 && scalar(N(sl).inputs[1], start)) ok = start == (a == th ? 0 : H);
          if (sl >= 0 && N(sl).op == "Split") ok = N(sl).outputs.size() == 2 && N(a).inputs[0] == N(sl).outputs[a == th ? 0 : 1];
          if (!ok) return fail(label(a) + ": must take the " + (a == th ? "first" : "second") + " half of the in_layer output (found " + label(sl) + ", start " + std::to_string(start) + ")");
          const int src = back(N(sl).inputs[0], {"Add"});
          if (src != nin) return fail(label(sl) + ": does not slice in_layers." + std::to_string(i) + " of " + Fl);
        }
      }
    }
    // ---- HiFi-GAN generator ----
    int npre;
    if (!check_conv("dec.conv_pre", false, 7, 1, 1, 3, 3, &npre)) return false;
    std::string stage_in = N(npre).outputs[0];  // what the next LeakyRelu → ConvTranspose reads
    for (int u = 0; u < c->n_ups; u++) {
      const int ku = c->up_kernels[u], su = c->up_rates[u];
      int nup;
      snprintf(p, sizeof p, "dec.ups.%d", u);
      if (!check_conv(p, true, ku, su, 1, (ku - su) / 2, (ku - su) / 2, &nup)) return false;
      const int lr = expect_lrelu(N(nup).inputs[0], 0.1, std::string(p));
      if (lr < 0) return false;
      if (N(lr).inputs[0] != stage_in) return fail(label(lr) + ": expected to read " + stage_in + " (the previous stage's output)");
      const std::string up_out = N(nup).outputs[0];
      std::vector<std::string> rb_out;
      for (int j = 0; j < c->n_rb; j++) {
        const int rb = u * c->n_rb + j, kk = c->rb_kernels[j];
        std::string xin = up_out;
        for (int di = 0; di < c->rb_n_dil; di++) {
          const int dil = c->rb_dilations[j][di];
          int nc1, nc2 = -1;
          if (c->resblock_type == 1) {
            snprintf(p, sizeof p, "dec.resblocks.%d.convs1.%d", rb, di);
            if (!check_conv(p, false, kk, 1, dil, (kk * dil - dil) / 2, (kk * dil - dil) / 2, &nc1)) return false;
            snprintf(p, sizeof p, "dec.resblocks.%d.convs2.%d", rb, di);
            if (!check_conv(p, false, kk, 1, 1, (kk - 1) / 2, (kk - 1) / 2, &nc2)) return false;
          } else {
            snprintf(p, sizeof p, "dec.resblocks.%d.convs.%d", rb, di);
            if (!check_conv(p, false, kk, 1, dil, (kk * dil - dil) / 2, (kk * dil - dil) / 2, &nc1)) return false;
          }
          const int l1 = expect_lrelu(N(nc1).inputs[0], 0.1, label(nc1));
          if (l1 < 0) return false;
          if (N(l1).inputs[0] != xin) return fail(label(l1) + ": expected to read " + xin + " (the ResBlock's running x)");
          int last = nc1;
          if (nc2 >= 0) {
            const int l2 = expect_lrelu(N(nc2).inputs[0], 0.1, label(nc2));
            if (l2 < 0) return false;
            if (back(N(l2).inputs[0]) != nc1) return fail(label(l2) + ": does not take convs1." + std::to_string(di));
            last = nc2;
          }
          std::vector<int> cs;
          fwd(N(last).outputs[0], cs);
          int radd = -1;
          for (int ci : cs)
            if (N(ci).op == "Add" && N(ci).inputs.size() == 2 && (N(ci).inputs[0] == xin || N(ci).inputs[1] == xin)) radd = ci;
          if (radd < 0) return fail(label(last) + ": its output is not added to the ResBlock's running x (" + xin + "); consumers: " + (cs.empty() ? "none" : label(cs[0])));
          xin = N(radd).outputs[0];
        }
        rb_out.push_back(xin);
      }
      // MRF: (r1 + r2 + …) / n_rb
      int cur = -1;
      {
        std::vector<int> cs;
        fwd(rb_out[0], cs);
        for (int ci : cs)
          if (N(ci).op == "Add") cur = ci;
        if (c->n_rb == 1) cur = -2;
      }
      std::string sum_t = rb_out[0];
      for (int j = 1; j < c->n_rb; j++) {
        std::vector<int> cs;
        fwd(sum_t, cs);
        int a = -1;
        for (int ci : cs)
          if (N(ci).op == "Add" && N(ci).inputs.size() == 2 && (N(ci).inputs[0] == rb_out[(size_t)j] || N(ci).inputs[1] == rb_out[(size_t)j])) a = ci;
        if (a < 0) return fail("stage " + std::to_string(u) + ": ResBlock " + std::to_string(j) + "'s output is not summed with the others (multi-receptive-field fusion)");
        sum_t = N(a).outputs[0];
      }
      (void)cur;
      {
        std::vector<int> cs;
        fwd(sum_t, cs);
        int dv = -1;
        double s = 0;
        for (int ci : cs) {
          const Node& n = N(ci);
          if (n.inputs.size() == 2 && scalar(n.inputs[1], s) && ((n.op == "Div" && close(s, c->n_rb)) || (n.op == "Mul" && close(s, 1.0 / c->n_rb)))) dv = ci;
          else if (n.op == "Div" || n.op == "Mul") return fail(label(ci) + ": the ResBlock mean divides by " + std::to_string(s) + ", the schedule by " + std::to_string(c->n_rb));
        }
        if (dv < 0) return fail("stage " + std::to_string(u) + ": the ResBlock sum is not divided by " + std::to_string(c->n_rb));
        stage_in = N(dv).outputs[0];
      }
    }
    {
      int npost;
      if (!check_conv("dec.conv_post", false, 7, 1, 1, 3, 3, &npost)) return false;
      const int lr = expect_lrelu(N(npost).inputs[0], 0.01, "dec.conv_post");
      if (lr < 0) return false;
      if (N(lr).inputs[0] != stage_in) return fail(label(lr) + ": expected to read " + stage_in);
      std::vector<int> cs;
      fwd(N(npost).outputs[0], cs);
      int th = -1;
      for (int ci : cs)
        if (N(ci).op == "Tanh") th = ci;
      if (th < 0) return fail(label(npost) + ": its output does not go through Tanh (found " + (cs.empty() ? std::string("no consumer") : label(cs[0])) + ")");
      std::string out = N(th).outputs[0];
      for (int hop = 0; hop < 4 && out != "output"; hop++) {
        auto it = cons.find(out);
        if (it == cons.end() || it->second.empty() || !in(N(it->second[0]).op, {"Identity", "Unsqueeze", "Squeeze", "Reshape"})) break;
        out = N(it->second[0]).outputs[0];
      }
      if (out != "output") return fail(label(th) + ": does not produce the graph output");
    }
    return true;
  }
};

}  // namespace

/* Refuses (PIPER_HIP_ERR_UNSUPPORTED, message = the first difference and the node it was found at) a graph that is not the computation
 * the launch schedule performs for `cfg`. Role of the reference's node-by-node execution (GraphExecutor.swift:227-265). */
PH_EXPORT int piper_hip_onnx_verify_graph(const piper_hip_onnx* m, const piper_hip_voice_config* cfg) {
  if (!m || !cfg) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx_verify_graph: null argument");
  Verifier v{m, cfg};
  if (!v.run()) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "onnx graph does not match the launch schedule: %s", v.why.c_str());
  return PIPER_HIP_OK;
}
