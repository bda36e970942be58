// onnx_model.h — the in-memory form of a parsed `.onnx` shared by the loader (onnx_loader.cpp) and the graph verifier (onnx_verify.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace ph {
namespace onnx {

struct Tensor {
  std::string name;
  int dtype = 0;  // ONNX TensorProto.DataType: 1 = FLOAT, 7 = INT64
  std::vector<int64_t> dims;
  const uint8_t* raw = nullptr;  size_t raw_len = 0;    // field 9
  const uint8_t* fdat = nullptr; size_t fdat_len = 0;   // field 4, packed
  std::vector<float> floats_unpacked;                    // field 4, unpacked encoding (rare)
  std::vector<int64_t> i64_unpacked;                     // field 7 (int64_data)
  int64_t count() const {  // -1 for negative dims or a product that does not fit (hostile / corrupt files)
    int64_t n = 1;
    for (int64_t d : dims) {
      if (d < 0 || (d != 0 && n > INT64_MAX / d)) return -1;
      n *= d;
    }
    return n;
  }
};

struct ConvNode {  // Conv / ConvTranspose with an initializer as weight
  std::string op, weight, bias, name;
  int64_t stride = 1, dilation = 1, group = 1, pad_l = 0, pad_r = 0;
};

// One NodeProto (ONNXLoader.swift:170-223: 1 input, 2 output, 3 name, 4 op_type, 5 attribute) with the attributes the verifier looks at.
struct Attr {
  std::string name;
  bool has_i = false, has_f = false, has_t = false;
  int64_t i = 0;
  float f = 0.0f;
  std::vector<int64_t> ints;
  std::string s;
  Tensor t;  // AttributeProto.t (Constant nodes)
};
struct Node {
  std::string op, name;
  std::vector<std::string> inputs, outputs;
  std::vector<Attr> attrs;
  const Attr* attr(const char* n) const {
    for (const Attr& a : attrs)
      if (a.name == n) return &a;
    return nullptr;
  }
};

}  // namespace onnx
}  // namespace ph

struct piper_hip_onnx {
  std::vector<uint8_t> owned;          // open_memory copy
  const uint8_t* data = nullptr;
  size_t size = 0;
  void* map = nullptr;
  size_t map_len = 0;
  int64_t ir_version = 0, opset = 0;
  int n_nodes = 0;
  std::vector<ph::onnx::Tensor> tensors;
  std::map<std::string, int> by_name;
  std::map<std::string, ph::onnx::ConvNode> conv_by_weight;
  std::map<std::string, ph::onnx::ConvNode> conv_by_node;  // by NodeProto.name ("/flow/flows.0/enc/in_layers.0/Conv")
  std::vector<ph::onnx::Node> nodes;                  // the whole graph, in file order
  std::vector<std::string> graph_inputs, graph_outputs;  // GraphProto 11 / 12 (ValueInfoProto.name), initializers excluded by the verifier
};

