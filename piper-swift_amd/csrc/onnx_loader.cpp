// onnx_loader.cpp — reads a Piper VITS `.onnx` (protobuf wire subset) and turns it into the packed voice blob.
//
// Role of Sources/PiperONNX (ONNXLoader.swift:25-387, ONNXIR.swift) for THIS library: SURVEY.md §8f row 1. Only what the
// hot path needs is decoded — ModelProto{1 ir_version, 7 graph, 8 opset_import}, GraphProto{1 node, 5 initializer},
// NodeProto{1 input, 4 op_type, 5 attribute}, AttributeProto{1 name, 3 i, 8 ints}, TensorProto{1 dims, 2 data_type,
// 4 float_data, 8 name, 9 raw_data} (the same field numbers the reference lists at ONNXLoader.swift:34-37, 94-99, 170-176,
// 214-223, 321-327); every other field is skipped by wire type. Host-only: no GPU is touched here.
//
// From the initializer shapes and the Conv / ConvTranspose attributes it infers the voice geometry
// (piper_hip_voice_config) and emits the initializers in the order include/piper_hip_voice_layout.h fixes, folding
// weight-norm pairs (weight_g, weight_v) if an export left them in.
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>

#include "../../include/piper_hip_voice_layout.h"
#include "common.h"

using namespace ph;

namespace {

struct Reader {  // protobuf wire format
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool at_end() const { return p >= end || !ok; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    ok = false;
    return 0;
  }
  Reader sub() {  // length-delimited payload
    const uint64_t n = varint();
    if (!ok || n > (uint64_t)(end - p)) { ok = false; return Reader{p, p}; }
    Reader r{p, p + n};
    p += n;
    return r;
  }
  void skip(int wire) {
    switch (wire) {
      case 0: (void)varint(); break;
      case 1: if (end - p >= 8) p += 8; else ok = false; break;
      case 2: (void)sub(); break;
      case 5: if (end - p >= 4) p += 4; else ok = false; break;
      default: ok = false;
    }
  }
};

}  // namespace

#include "onnx_model.h"
using namespace ph::onnx;

namespace {

bool parse_tensor(Reader r, Tensor& t) {
  while (!r.at_end()) {
    const uint64_t tag = r.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field == 1 && wire == 2) {  // dims, packed
      Reader d = r.sub();
      while (!d.at_end()) t.dims.push_back((int64_t)d.varint());
    } else if (field == 1 && wire == 0) {
      t.dims.push_back((int64_t)r.varint());
    } else if (field == 2 && wire == 0) {
      t.dtype = (int)r.varint();
    } else if (field == 4 && wire == 2) {
      Reader d = r.sub();
      t.fdat = d.p; t.fdat_len = (size_t)(d.end - d.p);
    } else if (field == 4 && wire == 5) {
      if (r.end - r.p < 4) return false;
      float f; memcpy(&f, r.p, 4); r.p += 4;
      t.floats_unpacked.push_back(f);
    } else if (field == 7 && wire == 2) {  // int64_data, packed (Constant tensors)
      Reader d = r.sub();
      while (!d.at_end()) t.i64_unpacked.push_back((int64_t)d.varint());
    } else if (field == 7 && wire == 0) {
      t.i64_unpacked.push_back((int64_t)r.varint());
    } else if (field == 8 && wire == 2) {
      Reader d = r.sub();
      t.name.assign((const char*)d.p, (size_t)(d.end - d.p));
    } else if (field == 9 && wire == 2) {
      Reader d = r.sub();
      t.raw = d.p; t.raw_len = (size_t)(d.end - d.p);
    } else {
      r.skip(wire);
    }
  }
  return r.ok;
}

bool parse_node(Reader r, piper_hip_onnx* m) {
  Node nd;
  ConvNode c;
  while (!r.at_end()) {
    const uint64_t tag = r.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field == 1 && wire == 2) {
      Reader d = r.sub();
      nd.inputs.emplace_back((const char*)d.p, (size_t)(d.end - d.p));
    } else if (field == 2 && wire == 2) {
      Reader d = r.sub();
      nd.outputs.emplace_back((const char*)d.p, (size_t)(d.end - d.p));
    } else if (field == 3 && wire == 2) {
      Reader d = r.sub();
      nd.name.assign((const char*)d.p, (size_t)(d.end - d.p));
    } else if (field == 4 && wire == 2) {
      Reader d = r.sub();
      nd.op.assign((const char*)d.p, (size_t)(d.end - d.p));
    } else if (field == 5 && wire == 2) {  // AttributeProto: 1 name, 2 f, 3 i, 4 s, 5 t, 8 ints (ONNXLoader.swift:214-223)
      Reader a = r.sub();
      Attr at;
      while (!a.at_end()) {
        const uint64_t t2 = a.varint();
        const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
        if (f2 == 1 && w2 == 2) { Reader d = a.sub(); at.name.assign((const char*)d.p, (size_t)(d.end - d.p)); }
        else if (f2 == 2 && w2 == 5) { if (a.end - a.p < 4) return false; memcpy(&at.f, a.p, 4); a.p += 4; at.has_f = true; }
        else if (f2 == 3 && w2 == 0) { at.i = (int64_t)a.varint(); at.has_i = true; }
        else if (f2 == 4 && w2 == 2) { Reader d = a.sub(); at.s.assign((const char*)d.p, (size_t)(d.end - d.p)); }
        else if (f2 == 5 && w2 == 2) { if (!parse_tensor(a.sub(), at.t)) return false; at.has_t = true; }
        else if (f2 == 8 && w2 == 2) { Reader d = a.sub(); while (!d.at_end()) at.ints.push_back((int64_t)d.varint()); }
        else if (f2 == 8 && w2 == 0) at.ints.push_back((int64_t)a.varint());
        else a.skip(w2);
      }
      if (!a.ok) return false;
      if (at.name == "strides" && !at.ints.empty()) c.stride = at.ints[0];
      else if (at.name == "dilations" && !at.ints.empty()) c.dilation = at.ints[0];
      else if (at.name == "group" && at.has_i) c.group = at.i;
      else if (at.name == "pads" && at.ints.size() >= 2) { c.pad_l = at.ints[0]; c.pad_r = at.ints[at.ints.size() / 2]; }
      nd.attrs.push_back(std::move(at));
    } else {
      r.skip(wire);
    }
  }
  if (!r.ok) return false;
  m->n_nodes++;
  if ((nd.op == "Conv" || nd.op == "ConvTranspose") && nd.inputs.size() >= 2) {
    c.op = nd.op;
    c.weight = nd.inputs[1];
    if (nd.inputs.size() >= 3) c.bias = nd.inputs[2];
    c.name = nd.name;
    m->conv_by_weight[c.weight] = c;
    if (!nd.name.empty()) m->conv_by_node[nd.name] = c;
  }
  m->nodes.push_back(std::move(nd));
  return true;
}

// ValueInfoProto{1 name}: graph inputs / outputs (GraphProto 11 / 12, ONNXLoader.swift:94-99)
std::string value_info_name(Reader r) {
  std::string name;
  while (!r.at_end()) {
    const uint64_t tag = r.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 2) { Reader d = r.sub(); name.assign((const char*)d.p, (size_t)(d.end - d.p)); }
    else r.skip((int)(tag & 7));
  }
  return name;
}

int parse_model(piper_hip_onnx* m) {
  Reader r{m->data, m->data + m->size};
  bool have_graph = false;
  while (!r.at_end()) {
    const uint64_t tag = r.varint();
    const int field = (int)(tag >> 3), wire = (int)(tag & 7);
    if (field == 1 && wire == 0) {
      m->ir_version = (int64_t)r.varint();
    } else if (field == 8 && wire == 2) {  // OperatorSetIdProto{1 domain, 2 version}: keep the default-domain version
      Reader o = r.sub();
      std::string domain;
      int64_t ver = 0;
      while (!o.at_end()) {
        const uint64_t t2 = o.varint();
        if ((t2 >> 3) == 1 && (t2 & 7) == 2) { Reader d = o.sub(); domain.assign((const char*)d.p, (size_t)(d.end - d.p)); }
        else if ((t2 >> 3) == 2 && (t2 & 7) == 0) ver = (int64_t)o.varint();
        else o.skip((int)(t2 & 7));
      }
      if (domain.empty() || domain == "ai.onnx") m->opset = ver;
    } else if (field == 7 && wire == 2) {
      Reader g = r.sub();
      have_graph = true;
      while (!g.at_end()) {
        const uint64_t t2 = g.varint();
        const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
        if (f2 == 1 && w2 == 2) {
          if (!parse_node(g.sub(), m)) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: malformed NodeProto");
        } else if (f2 == 5 && w2 == 2) {
          Tensor t;
          if (!parse_tensor(g.sub(), t)) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: malformed TensorProto");
          m->by_name[t.name] = (int)m->tensors.size();
          m->tensors.push_back(std::move(t));
        } else if (f2 == 11 && w2 == 2) {
          m->graph_inputs.push_back(value_info_name(g.sub()));
        } else if (f2 == 12 && w2 == 2) {
          m->graph_outputs.push_back(value_info_name(g.sub()));
        } else {
          g.skip(w2);
        }
      }
      if (!g.ok) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: malformed GraphProto");
    } else {
      r.skip(wire);
    }
  }
  if (!r.ok) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: malformed ModelProto");
  if (!have_graph) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: no graph in model");
  return PIPER_HIP_OK;
}

const Tensor* find(const piper_hip_onnx* m, const std::string& name) {
  auto it = m->by_name.find(name);
  return it == m->by_name.end() ? nullptr : &m->tensors[it->second];
}

// Module path of a layout name → the scope name torch.onnx gives the module's node: attribute accesses are joined by '/',
// ModuleList indices stay glued to their list with '.' ("flow.flows.0.enc.in_layers.1" → "/flow/flows.0/enc/in_layers.1";
// the convention behind the reference's own "/enc_p/encoder/attn_layers." node-name match, GraphExecutor.swift:908).
std::string scope_of(const std::string& module) {
  std::string out;
  size_t i = 0;
  while (i < module.size()) {
    size_t j = module.find('.', i);
    if (j == std::string::npos) j = module.size();
    const std::string part = module.substr(i, j - i);
    const bool numeric = !part.empty() && part.find_first_not_of("0123456789") == std::string::npos;
    out += (numeric && !out.empty()) ? "." : "/";
    out += part;
    i = j + 1;
  }
  return out;
}

// The Conv / ConvTranspose node of module `module`, if the export kept scope names.
const ConvNode* node_of(const piper_hip_onnx* m, const std::string& module) {
  const std::string sc = scope_of(module);
  auto it = m->conv_by_node.find(sc + "/Conv");
  if (it == m->conv_by_node.end()) it = m->conv_by_node.find(sc + "/ConvTranspose");
  return it == m->conv_by_node.end() ? nullptr : &it->second;
}

// Initializer behind a layout name: by its module-path name; else through the module's node — weight-norm parametrised
// layers (flow WaveNet, HiFi-GAN) are constant-folded by the exporter into anonymous `onnx::Conv_1234` initializers that
// only the node's input[1] / input[2] still identify.
const Tensor* resolve(const piper_hip_onnx* m, const std::string& name) {
  if (const Tensor* t = find(m, name)) return t;
  const bool is_w = name.size() > 7 && name.compare(name.size() - 7, 7, ".weight") == 0;
  const bool is_b = name.size() > 5 && name.compare(name.size() - 5, 5, ".bias") == 0;
  if (!is_w && !is_b) return nullptr;
  const ConvNode* n = node_of(m, name.substr(0, name.size() - (is_w ? 7 : 5)));
  if (!n) return nullptr;
  return find(m, is_w ? n->weight : n->bias);
}

// strides / dilations of the node that consumes a layout weight (by initializer name, then by scope name)
const ConvNode* conv_node_for(const piper_hip_onnx* m, const std::string& weight_name) {
  auto it = m->conv_by_weight.find(weight_name);
  if (it != m->conv_by_weight.end()) return &it->second;
  it = m->conv_by_weight.find(weight_name + "_v");
  if (it != m->conv_by_weight.end()) return &it->second;
  if (weight_name.size() > 7) return node_of(m, weight_name.substr(0, weight_name.size() - 7));
  return nullptr;
}

// floats of a FLOAT tensor (raw little-endian or float_data), TensorValue.swift:45-116
int read_floats(const Tensor& t, float* dst, size_t n) {
  if (t.dtype != 1) PH_FAIL(PIPER_HIP_ERR_TYPE, "onnx: initializer '%s' has data_type %d, expected FLOAT(1)", t.name.c_str(), t.dtype);
  if ((int64_t)n != t.count()) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: initializer '%s' has %lld elements, expected %zu", t.name.c_str(), (long long)t.count(), n);
  if (t.raw_len) {
    if (t.raw_len != n * 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: initializer '%s' raw_data is %zu bytes, expected %zu", t.name.c_str(), t.raw_len, n * 4);
    memcpy(dst, t.raw, n * 4);
  } else if (t.fdat_len) {
    if (t.fdat_len != n * 4) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: initializer '%s' float_data is %zu bytes, expected %zu", t.name.c_str(), t.fdat_len, n * 4);
    memcpy(dst, t.fdat, n * 4);
  } else if (t.floats_unpacked.size() == n) {
    memcpy(dst, t.floats_unpacked.data(), n * 4);
  } else if (n != 0) {
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: initializer '%s' carries no data (external data is not supported)", t.name.c_str());
  }
  return PIPER_HIP_OK;
}

std::string fmt(const char* f, int a, int b = 0) {
  char buf[160];
  snprintf(buf, sizeof buf, f, a, b);
  return buf;
}

struct BlobFill {
  const piper_hip_onnx* m;
  float* blob;
  int rc;
  std::vector<float> g, v;
};

void fill_visit(const piper_tensor_desc* d, void* user) {
  BlobFill* b = (BlobFill*)user;
  if (b->rc) return;
  const std::string name = d->name;
  float* dst = b->blob + d->offset;
  // a weight-norm pair left in the export takes precedence over whatever the module's node lists as its weight input
  const bool wn_pair = !find(b->m, name) && find(b->m, name + "_g") && find(b->m, name + "_v");
  if (const Tensor* t = wn_pair ? nullptr : resolve(b->m, name)) {
    // shape check: same element count and, when ranks agree, the same dims
    if ((int)t->dims.size() == d->rank)
      for (int i = 0; i < d->rank; i++)
        if (t->dims[i] != d->shape[i]) {
          set_error("onnx: initializer '%s' dim %d is %lld, the voice geometry expects %lld", d->name, i, (long long)t->dims[i], d->shape[i]);
          b->rc = PIPER_HIP_ERR_SHAPE;
          return;
        }
    b->rc = read_floats(*t, dst, d->count);
    return;
  }
  // weight norm left in the export: w = g · v / ‖v‖ over everything but dim 0
  const Tensor* tg = find(b->m, name + "_g");
  const Tensor* tv = find(b->m, name + "_v");
  if (tg && tv && d->shape[0] > 0 && tg->count() == d->shape[0]) {
    b->g.resize((size_t)d->shape[0]);
    b->v.resize(d->count);
    if ((b->rc = read_floats(*tg, b->g.data(), b->g.size()))) return;
    if ((b->rc = read_floats(*tv, b->v.data(), d->count))) return;
    const size_t per = d->count / (size_t)d->shape[0];
    for (long long r = 0; r < d->shape[0]; r++) {
      double ss = 0.0;
      for (size_t i = 0; i < per; i++) ss += (double)b->v[r * per + i] * b->v[r * per + i];
      const float scale = (float)(b->g[r] / std::sqrt(ss));
      for (size_t i = 0; i < per; i++) dst[r * per + i] = b->v[r * per + i] * scale;
    }
    return;
  }
  set_error("onnx: initializer '%s' not found (nor %s_g/%s_v)", d->name, d->name, d->name);
  b->rc = PIPER_HIP_ERR_ARG;
}

}  // namespace

PH_EXPORT int piper_hip_onnx_open_memory(const void* data, size_t size, piper_hip_onnx** out) {
  if (!data || !out) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx_open_memory: null argument");
  std::unique_ptr<piper_hip_onnx> m(new piper_hip_onnx());
  m->owned.assign((const uint8_t*)data, (const uint8_t*)data + size);
  m->data = m->owned.data();
  m->size = size;
  int rc = parse_model(m.get());
  if (rc) return rc;
  *out = m.release();
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_onnx_open(const char* path, piper_hip_onnx** out) {
  if (!path || !out) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx_open: null argument");
  const int fd = open(path, O_RDONLY);
  if (fd < 0) PH_FAIL(PIPER_HIP_ERR_ARG, "onnx_open: cannot open '%s'", path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); PH_FAIL(PIPER_HIP_ERR_ARG, "onnx_open: cannot stat '%s'", path); }
  void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);  // 60–110 MB voices: map, do not copy
  close(fd);
  if (map == MAP_FAILED) PH_FAIL(PIPER_HIP_ERR_ALLOC, "onnx_open: mmap of '%s' failed", path);
  std::unique_ptr<piper_hip_onnx> m(new piper_hip_onnx());
  m->map = map; m->map_len = (size_t)st.st_size;
  m->data = (const uint8_t*)map; m->size = (size_t)st.st_size;
  int rc = parse_model(m.get());
  if (rc) { munmap(map, m->map_len); return rc; }
  *out = m.release();
  return PIPER_HIP_OK;
}

PH_EXPORT void piper_hip_onnx_close(piper_hip_onnx* m) {
  if (!m) return;
  if (m->map) munmap(m->map, m->map_len);
  delete m;
}

PH_EXPORT int piper_hip_onnx_counts(const piper_hip_onnx* m, int64_t* ir_version, int64_t* opset, int* n_nodes, int* n_initializers) {
  if (!m) PH_FAIL(PIPER_HIP_ERR_ARG, "null model");
  if (ir_version) *ir_version = m->ir_version;
  if (opset) *opset = m->opset;
  if (n_nodes) *n_nodes = m->n_nodes;
  if (n_initializers) *n_initializers = (int)m->tensors.size();
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_onnx_initializer(const piper_hip_onnx* m, int index, piper_hip_onnx_tensor_info* out) {
  if (!m || !out) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (index < 0 || index >= (int)m->tensors.size()) PH_FAIL(PIPER_HIP_ERR_ARG, "initializer index %d out of range", index);
  const Tensor& t = m->tensors[index];
  memset(out, 0, sizeof *out);
  snprintf(out->name, sizeof out->name, "%s", t.name.c_str());
  out->data_type = t.dtype;
  out->rank = (int32_t)t.dims.size();
  for (int i = 0; i < out->rank && i < 8; i++) out->dims[i] = t.dims[i];
  out->count = t.count();
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_onnx_find(const piper_hip_onnx* m, const char* name) {
  if (!m || !name) return -1;
  auto it = m->by_name.find(name);
  return it == m->by_name.end() ? -1 : it->second;
}

PH_EXPORT int piper_hip_onnx_read_f32(const piper_hip_onnx* m, int index, float* dst, size_t n) {
  if (!m || !dst) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (index < 0 || index >= (int)m->tensors.size()) PH_FAIL(PIPER_HIP_ERR_ARG, "initializer index %d out of range", index);
  return read_floats(m->tensors[index], dst, n);
}

PH_EXPORT int piper_hip_onnx_infer_config(const piper_hip_onnx* m, piper_hip_voice_config* cfg) {
  if (!m || !cfg) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  memset(cfg, 0, sizeof *cfg);
  // Multi-speaker voices condition the flow, the duration predictor and the generator on a speaker embedding `g`
  // (emb_g + cond / cond_layer convs). None of that is implemented here, and ignoring it would synthesise the wrong voice
  // without any error — refuse instead.
  for (const Tensor& it : m->tensors)
    if (it.name.compare(0, 5, "emb_g") == 0 || it.name.find(".cond.") != std::string::npos || it.name.find(".cond_layer.") != std::string::npos)
      PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "onnx: multi-speaker voice (initializer '%s'): speaker conditioning is not implemented", it.name.c_str());
  for (const Tensor& it : m->tensors)  // geometry fields are int32: refuse dims that would be truncated
    for (int64_t dm : it.dims)
      if (dm < 0 || dm > INT32_MAX) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: initializer '%s' has a dimension of %lld", it.name.c_str(), (long long)dm);
  auto need = [&](const std::string& name, size_t rank) -> const Tensor* {
    const Tensor* t = resolve(m, name);
    if (!t) t = find(m, name + "_v");  // weight norm left in
    if (!t) { set_error("onnx: not a Piper VITS voice: initializer '%s' missing", name.c_str()); return nullptr; }
    if (t->dims.size() != rank) { set_error("onnx: initializer '%s' has rank %zu, expected %zu", name.c_str(), t->dims.size(), rank); return nullptr; }
    return t;
  };
  auto has = [&](const std::string& name) { return resolve(m, name) || find(m, name + "_v"); };
  const Tensor* t;
  if (!(t = need("enc_p.emb.weight", 2))) return PIPER_HIP_ERR_ARG;
  cfg->n_vocab = (int32_t)t->dims[0];
  cfg->hidden = (int32_t)t->dims[1];
  while (has(fmt("enc_p.encoder.attn_layers.%d.conv_q.weight", cfg->n_layers))) cfg->n_layers++;
  // [1, 2w+1, d] in a VITS export (one shared relative-position head); [2w+1, d] accepted too
  t = find(m, "enc_p.encoder.attn_layers.0.emb_rel_k");
  if (!t || t->dims.size() < 2 || t->dims.size() > 3)
    PH_FAIL(PIPER_HIP_ERR_ARG, "onnx: not a Piper VITS voice: 'enc_p.encoder.attn_layers.0.emb_rel_k' missing or not rank 2/3");
  {
    const int64_t rows = t->dims[t->dims.size() - 2], d = t->dims[t->dims.size() - 1];
    cfg->window = (int32_t)((rows - 1) / 2);
    cfg->n_heads = d > 0 ? (int32_t)(cfg->hidden / d) : 0;
  }
  if (!(t = need("enc_p.encoder.ffn_layers.0.conv_1.weight", 3))) return PIPER_HIP_ERR_ARG;
  cfg->ffn = (int32_t)t->dims[0];
  cfg->ffn_kernel = (int32_t)t->dims[2];
  if (!(t = need("enc_p.proj.weight", 3))) return PIPER_HIP_ERR_ARG;
  cfg->inter = (int32_t)(t->dims[0] / 2);
  while (has(fmt("flow.flows.%d.pre.weight", 2 * cfg->n_flows))) cfg->n_flows++;
  while (has(fmt("flow.flows.0.enc.in_layers.%d.weight", cfg->wn_layers))) cfg->wn_layers++;
  if (!(t = need("flow.flows.0.enc.in_layers.0.weight", 3))) return PIPER_HIP_ERR_ARG;
  cfg->wn_kernel = (int32_t)t->dims[2];
  if (!(t = need("dec.conv_pre.weight", 3))) return PIPER_HIP_ERR_ARG;
  cfg->up_initial = (int32_t)t->dims[0];
  while (cfg->n_ups < PIPER_HIP_MAX_UPS && has(fmt("dec.ups.%d.weight", cfg->n_ups))) {
    const std::string wn = fmt("dec.ups.%d.weight", cfg->n_ups);
    if (!(t = need(wn, 3))) return PIPER_HIP_ERR_ARG;
    cfg->up_kernels[cfg->n_ups] = (int32_t)t->dims[2];
    const ConvNode* cn = conv_node_for(m, wn);
    // the stride lives on the ConvTranspose node; HiFi-GAN's convention K = 2·stride is the fallback
    const int64_t st = cn ? cn->stride : t->dims[2] / 2;
    if (st < 1 || st > 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: upsampler %d has stride %lld", cfg->n_ups, (long long)st);
    cfg->up_rates[cfg->n_ups] = (int32_t)st;
    cfg->n_ups++;
  }
  if (has(fmt("dec.ups.%d.weight", cfg->n_ups))) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "onnx: more than %d upsampling stages", PIPER_HIP_MAX_UPS);
  cfg->resblock_type = has("dec.resblocks.0.convs1.0.weight") ? 1 : 2;
  const char* first = cfg->resblock_type == 1 ? "dec.resblocks.%d.convs1.%d.weight" : "dec.resblocks.%d.convs.%d.weight";
  int n_resblocks = 0;
  while (has(fmt(first, n_resblocks, 0))) n_resblocks++;
  if (cfg->n_ups <= 0 || n_resblocks % cfg->n_ups) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: %d resblocks over %d stages", n_resblocks, cfg->n_ups);
  cfg->n_rb = n_resblocks / cfg->n_ups;
  if (cfg->n_rb > PIPER_HIP_MAX_RB) PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "onnx: %d resblocks per stage (max %d)", cfg->n_rb, PIPER_HIP_MAX_RB);
  while (cfg->rb_n_dil < 3 && has(fmt(first, 0, cfg->rb_n_dil))) cfg->rb_n_dil++;
  for (int j = 0; j < cfg->n_rb; j++) {
    if (!(t = need(fmt(first, j, 0), 3))) return PIPER_HIP_ERR_ARG;
    cfg->rb_kernels[j] = (int32_t)t->dims[2];
    for (int d = 0; d < cfg->rb_n_dil; d++) {
      const ConvNode* cn = conv_node_for(m, fmt(first, j, d));
      static const int fallback[3] = {1, 3, 5};
      const int64_t dl = cn ? cn->dilation : fallback[d];
      if (dl < 1 || dl > 64) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx: resblock %d conv %d has dilation %lld", j, d, (long long)dl);
      cfg->rb_dilations[j][d] = (int32_t)dl;
    }
  }
  // stochastic duration predictor (absent ⇒ dp_present = 0: durations must then be supplied per utterance)
  if (has("dp.pre.weight") && has("dp.flows.0.m")) {
    cfg->dp_present = 1;
    if (!(t = need("dp.convs.convs_sep.0.weight", 3))) return PIPER_HIP_ERR_ARG;
    cfg->dp_kernel = (int32_t)t->dims[2];
    while (cfg->dp_dds_layers < 8 && has(fmt("dp.convs.convs_sep.%d.weight", cfg->dp_dds_layers))) cfg->dp_dds_layers++;
    int used = 0;  // ConvFlows present: module indices 3, 5, … (flow 1 is dropped by the reverse pass, so not exported)
    while (used < 16 && has(fmt("dp.flows.%d.pre.weight", 2 * used + 3))) used++;
    cfg->dp_n_flows = used + 1;
    if (!(t = need("dp.flows.3.proj.weight", 3))) return PIPER_HIP_ERR_ARG;
    cfg->dp_bins = (int32_t)((t->dims[0] + 1) / 3);
    cfg->dp_tail_bound = 5.0f;  // a constant folded into the graph, not an initializer: VITS' fixed value
  }
  cfg->sample_rate = 22050;  // lives in the voice's .onnx.json (piper_hip_piper_json), not in the graph
  size_t n = 0;
  return piper_hip_voice_blob_floats(cfg, &n);  // validates the geometry
}

PH_EXPORT int piper_hip_onnx_build_blob(const piper_hip_onnx* m, const piper_hip_voice_config* cfg, float* host_blob, size_t n_floats) {
  if (!m || !cfg || !host_blob) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  // the weights are only handed out for a graph that IS the computation the schedule performs (onnx_verify.cpp)
  const int vrc = piper_hip_onnx_verify_graph(m, cfg);
  if (vrc) return vrc;
  return piper_hip_onnx_build_blob_unchecked(m, cfg, host_blob, n_floats);
}

PH_EXPORT int piper_hip_onnx_build_blob_unchecked(const piper_hip_onnx* m, const piper_hip_voice_config* cfg, float* host_blob, size_t n_floats) {
  if (!m || !cfg || !host_blob) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  size_t need = 0;
  int rc = piper_hip_voice_blob_floats(cfg, &need);
  if (rc) return rc;
  if (n_floats != need) PH_FAIL(PIPER_HIP_ERR_SHAPE, "onnx_build_blob: blob has %zu floats, the geometry needs %zu", n_floats, need);
  BlobFill b{m, host_blob, PIPER_HIP_OK, {}, {}};
  piper_hip_layout_walk(cfg, fill_visit, &b);
  return b.rc;
}

// ---- the voice's `.onnx.json` (PiperConfig.swift:3-47): the few numbers this library needs, by key
namespace {
bool json_number(const std::string& s, const char* key, double* out, size_t from = 0, size_t* at = nullptr) {
  const std::string k = std::string("\"") + key + "\"";
  size_t p = s.find(k, from);
  if (p == std::string::npos) return false;
  p = s.find(':', p + k.size());
  if (p == std::string::npos) return false;
  char* endp = nullptr;
  const double v = strtod(s.c_str() + p + 1, &endp);
  if (endp == s.c_str() + p + 1) return false;
  *out = v;
  if (at) *at = p;
  return true;
}
}  // namespace

PH_EXPORT int piper_hip_piper_json(const char* json_text, piper_hip_piper_json_info* out) {
  if (!json_text || !out) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  const std::string s(json_text);
  memset(out, 0, sizeof *out);
  double v = 0;
  const size_t audio = s.find("\"audio\"");
  if (audio == std::string::npos || !json_number(s, "sample_rate", &v, audio)) PH_FAIL(PIPER_HIP_ERR_ARG, "piper json: audio.sample_rate missing");
  out->sample_rate = (int32_t)v;
  if (!json_number(s, "num_symbols", &v)) PH_FAIL(PIPER_HIP_ERR_ARG, "piper json: num_symbols missing");
  out->num_symbols = (int32_t)v;
  out->num_speakers = json_number(s, "num_speakers", &v) ? (int32_t)v : 1;
  const size_t inf = s.find("\"inference\"");
  out->noise_scale = 0.667f; out->length_scale = 1.0f; out->noise_w = 0.8f;  // Piper defaults
  if (inf != std::string::npos) {
    if (json_number(s, "noise_scale", &v, inf)) out->noise_scale = (float)v;
    if (json_number(s, "length_scale", &v, inf)) out->length_scale = (float)v;
    if (json_number(s, "noise_w", &v, inf)) out->noise_w = (float)v;
  }
  return PIPER_HIP_OK;
}

PH_EXPORT int piper_hip_voice_check_json(const piper_hip_voice_config* cfg, const piper_hip_piper_json_info* info) {
  if (!cfg || !info) PH_FAIL(PIPER_HIP_ERR_ARG, "null argument");
  if (info->num_speakers > 1)
    PH_FAIL(PIPER_HIP_ERR_UNSUPPORTED, "voice has %d speakers: speaker conditioning (emb_g / cond layers) is not implemented", info->num_speakers);
  if (info->num_symbols != cfg->n_vocab)
    PH_FAIL(PIPER_HIP_ERR_SHAPE, "voice json says %d symbols, the graph's embedding has %d rows", info->num_symbols, cfg->n_vocab);
  return PIPER_HIP_OK;
}
