"""A Caffe SSD face detector written by the tests in the public res10 style (pre-activation ResNet trunk with
BatchNorm/Scale/ReLU layers, in-place tops, a BatchNorm on the data blob, Normalize on the first source, the mbox
branch with Permute/Flatten/Concat/Reshape/Softmax, PriorBox and DetectionOutput), its random weights, and a writer of
the caffemodel wire format (fp32 `data`, or OpenCV's fp16 `raw_data`).  Topology and names are the tests' own: the
real deploy.prototxt is not in the reference tree."""
import struct

import numpy as np


def _conv(name, bottom, top, n, k, stride=1, pad=0, dil=1, bias=False):
    s = f'layer {{ name: "{name}" type: "Convolution" bottom: "{bottom}" top: "{top}"\n  convolution_param {{ num_output: {n} kernel_size: {k}'
    if stride != 1:
        s += f" stride: {stride}"
    if pad:
        s += f" pad: {pad}"
    if dil != 1:
        s += f" dilation: {dil}"
    if not bias:
        s += " bias_term: false"
    return s + " } }\n"


def _bn_scale_relu(prefix, blob, relu=True):
    s = (f'layer {{ name: "{prefix}_bn" type: "BatchNorm" bottom: "{blob}" top: "{blob}" batch_norm_param {{ eps: 1e-5 }} }}\n'
         f'layer {{ name: "{prefix}_scale" type: "Scale" bottom: "{blob}" top: "{blob}" scale_param {{ bias_term: true }} }}\n')
    if relu:
        s += f'layer {{ name: "{prefix}_relu" type: "ReLU" bottom: "{blob}" top: "{blob}" }}\n'
    return s


def build(seed=0, fp16=False):
    """-> (prototxt text, caffemodel bytes, blobs dict as a real parser must recover them)"""
    rs = np.random.RandomState(seed)
    blobs = {}
    txt = ['name: "testnet_res10_style"\ninput: "data"\ninput_shape { dim: 1 dim: 3 dim: 300 dim: 300 }\n']

    def conv(name, bottom, top, ci, n, k, stride=1, pad=0, dil=1, bias=False, gain=1.0):
        txt.append(_conv(name, bottom, top, n, k, stride, pad, dil, bias))
        blobs[name] = [(rs.randn(n, ci, k, k) * gain / np.sqrt(ci * k * k)).astype(np.float32)]
        if bias:
            blobs[name].append((rs.randn(n) * 0.05).astype(np.float32))

    def bn(prefix, blob, c, relu=True):
        txt.append(_bn_scale_relu(prefix, blob, relu))
        blobs[prefix + "_bn"] = [(rs.randn(c) * 0.1).astype(np.float32) * 2.0, rs.uniform(0.5, 1.5, c).astype(np.float32) * 2.0,
                                 np.asarray([2.0], np.float32)]                     # scale_factor 2: mean/var stored doubled
        blobs[prefix + "_scale"] = [rs.uniform(0.7, 1.3, c).astype(np.float32), (rs.randn(c) * 0.1).astype(np.float32)]

    # data BatchNorm (as res10's data_bn / data_scale), conv1 + BN + ReLU, pool
    txt.append(_bn_scale_relu("data", "data", relu=False).replace('"data_bn"', '"data_bn"'))
    blobs["data_bn"] = [(rs.randn(3) * 5).astype(np.float32), rs.uniform(2000, 4000, 3).astype(np.float32), np.asarray([1.0], np.float32)]
    blobs["data_scale"] = [rs.uniform(0.8, 1.2, 3).astype(np.float32), (rs.randn(3) * 0.1).astype(np.float32)]
    conv("conv1_h", "data", "conv1_h", 3, 32, 7, 2, 3, bias=True, gain=1.4)
    bn("conv1", "conv1_h", 32)
    txt.append('layer { name: "conv1_pool" type: "Pooling" bottom: "conv1_h" top: "conv1_pool" pooling_param { kernel_size: 3 stride: 2 } }\n')
    # block A (32 ch, identity shortcut): conv - bn - relu - conv ; sum with the block input
    conv("layer_64_1_conv1_h", "conv1_pool", "layer_64_1_conv1_h", 32, 32, 3, 1, 1, gain=1.4)
    bn("layer_64_1_2", "layer_64_1_conv1_h", 32)
    conv("layer_64_1_conv2_h", "layer_64_1_conv1_h", "layer_64_1_conv2_h", 32, 32, 3, 1, 1, gain=0.7)
    txt.append('layer { name: "layer_64_1_sum" type: "Eltwise" bottom: "layer_64_1_conv2_h" bottom: "conv1_pool" top: "layer_64_1_sum" }\n')

    def preact_block(tag, src, ci, co, stride, dil=1):
        # bn - scale - relu (NOT in place: the sum is also the shortcut's... here the shortcut reads the activated tensor)
        txt.append(f'layer {{ name: "{tag}_bn1" type: "BatchNorm" bottom: "{src}" top: "{tag}_bn1" }}\n'
                   f'layer {{ name: "{tag}_scale1" type: "Scale" bottom: "{tag}_bn1" top: "{tag}_bn1" scale_param {{ bias_term: true }} }}\n'
                   f'layer {{ name: "{tag}_relu1" type: "ReLU" bottom: "{tag}_bn1" top: "{tag}_bn1" }}\n')
        blobs[tag + "_bn1"] = [(rs.randn(ci) * 0.1).astype(np.float32), rs.uniform(0.5, 1.5, ci).astype(np.float32), np.asarray([1.0], np.float32)]
        blobs[tag + "_scale1"] = [rs.uniform(0.7, 1.3, ci).astype(np.float32), (rs.randn(ci) * 0.1).astype(np.float32)]
        conv(f"{tag}_conv1_h", f"{tag}_bn1", f"{tag}_conv1_h", ci, co, 3, stride, dil, dil, gain=1.4)
        bn(f"{tag}_2", f"{tag}_conv1_h", co)
        conv(f"{tag}_conv2_h", f"{tag}_conv1_h", f"{tag}_conv2_h", co, co, 3, 1, dil, dil, gain=0.7)
        conv(f"{tag}_conv_expand_h", f"{tag}_bn1", f"{tag}_conv_expand_h", ci, co, 1, stride, 0, gain=0.7)
        txt.append(f'layer {{ name: "{tag}_sum" type: "Eltwise" bottom: "{tag}_conv2_h" bottom: "{tag}_conv_expand_h" top: "{tag}_sum" }}\n')
        return f"{tag}_sum", f"{tag}_bn1"

    s128, bn128 = preact_block("layer_128_1", "layer_64_1_sum", 32, 128, 2)          # 38
    s256, bn256 = preact_block("layer_256_1", s128, 128, 256, 2)                     # 19; bn256 = activated 128-ch, 38x38 map
    s512, _ = preact_block("layer_512_1", s256, 256, 256, 1, dil=2)                  # 19, dilated
    txt.append('layer { name: "last_bn_h" type: "BatchNorm" bottom: "%s" top: "%s" }\n'
               'layer { name: "last_scale_h" type: "Scale" bottom: "%s" top: "%s" scale_param { bias_term: true } }\n'
               'layer { name: "last_relu" type: "ReLU" bottom: "%s" top: "fc7" }\n' % (s512, s512, s512, s512, s512))
    blobs["last_bn_h"] = [(rs.randn(256) * 0.1).astype(np.float32), rs.uniform(0.5, 1.5, 256).astype(np.float32), np.asarray([1.0], np.float32)]
    blobs["last_scale_h"] = [rs.uniform(0.7, 1.3, 256).astype(np.float32), (rs.randn(256) * 0.1).astype(np.float32)]

    def extra(name, src, ci, co, k, stride, pad):
        conv(name, src, name, ci, co, k, stride, pad, bias=True, gain=1.4)
        txt.append(f'layer {{ name: "{name}_relu" type: "ReLU" bottom: "{name}" top: "{name}" }}\n')

    extra("conv6_1_h", "fc7", 256, 128, 1, 1, 0)
    extra("conv6_2_h", "conv6_1_h", 128, 256, 3, 2, 1)      # 10
    extra("conv7_1_h", "conv6_2_h", 256, 64, 1, 1, 0)
    extra("conv7_2_h", "conv7_1_h", 64, 128, 3, 2, 1)       # 5
    extra("conv8_1_h", "conv7_2_h", 128, 64, 1, 1, 0)
    extra("conv8_2_h", "conv8_1_h", 64, 128, 3, 1, 0)       # 3
    extra("conv9_1_h", "conv8_2_h", 128, 64, 1, 1, 0)
    extra("conv9_2_h", "conv9_1_h", 64, 128, 3, 1, 0)       # 1
    txt.append('layer { name: "conv4_3_norm" type: "Normalize" bottom: "%s" top: "conv4_3_norm"\n'
               '  norm_param { across_spatial: false scale_filler { type: "constant" value: 20 } channel_shared: false } }\n' % bn256)
    blobs["conv4_3_norm"] = [rs.uniform(15, 25, 128).astype(np.float32)]
    srcs = [("conv4_3_norm", 128, 30, 60, [2], 8), ("fc7", 256, 60, 111, [2, 3], 16), ("conv6_2_h", 256, 111, 162, [2, 3], 32),
            ("conv7_2_h", 128, 162, 213, [2, 3], 64), ("conv8_2_h", 128, 213, 264, [2], 100), ("conv9_2_h", 128, 264, 315, [2], 300)]
    for s, c, mn, mx, ars, step in srcs:
        p = 2 + 2 * len(ars)
        conv(f"{s}_mbox_loc", s, f"{s}_mbox_loc", c, 4 * p, 3, 1, 1, bias=True, gain=0.5)
        txt.append(f'layer {{ name: "{s}_mbox_loc_perm" type: "Permute" bottom: "{s}_mbox_loc" top: "{s}_mbox_loc_perm" permute_param {{ order: 0 order: 2 order: 3 order: 1 }} }}\n'
                   f'layer {{ name: "{s}_mbox_loc_flat" type: "Flatten" bottom: "{s}_mbox_loc_perm" top: "{s}_mbox_loc_flat" flatten_param {{ axis: 1 }} }}\n')
        conv(f"{s}_mbox_conf", s, f"{s}_mbox_conf", c, 2 * p, 3, 1, 1, bias=True, gain=0.6)
        blobs[f"{s}_mbox_conf"][1][0::2] += 1.0                                   # background bias: sparse detections
        txt.append(f'layer {{ name: "{s}_mbox_conf_perm" type: "Permute" bottom: "{s}_mbox_conf" top: "{s}_mbox_conf_perm" permute_param {{ order: 0 order: 2 order: 3 order: 1 }} }}\n'
                   f'layer {{ name: "{s}_mbox_conf_flat" type: "Flatten" bottom: "{s}_mbox_conf_perm" top: "{s}_mbox_conf_flat" flatten_param {{ axis: 1 }} }}\n')
        ar = " ".join(f"aspect_ratio: {a}" for a in ars)
        txt.append(f'layer {{ name: "{s}_mbox_priorbox" type: "PriorBox" bottom: "{s}" bottom: "data" top: "{s}_mbox_priorbox"\n'
                   f'  prior_box_param {{ min_size: {mn} max_size: {mx} {ar} flip: true clip: false variance: 0.1 variance: 0.1 variance: 0.2 variance: 0.2 step: {step} offset: 0.5 }} }}\n')
    names = [s for s, *_ in srcs]
    cat = lambda suffix: " ".join(f'bottom: "{n}{suffix}"' for n in names)           # noqa: E731
    txt.append(f'layer {{ name: "mbox_loc" type: "Concat" {cat("_mbox_loc_flat")} top: "mbox_loc" concat_param {{ axis: 1 }} }}\n'
               f'layer {{ name: "mbox_conf" type: "Concat" {cat("_mbox_conf_flat")} top: "mbox_conf" concat_param {{ axis: 1 }} }}\n'
               f'layer {{ name: "mbox_priorbox" type: "Concat" {cat("_mbox_priorbox")} top: "mbox_priorbox" concat_param {{ axis: 2 }} }}\n'
               'layer { name: "mbox_conf_reshape" type: "Reshape" bottom: "mbox_conf" top: "mbox_conf_reshape" reshape_param { shape { dim: 0 dim: -1 dim: 2 } } }\n'
               'layer { name: "mbox_conf_softmax" type: "Softmax" bottom: "mbox_conf_reshape" top: "mbox_conf_softmax" softmax_param { axis: 2 } }\n'
               'layer { name: "mbox_conf_flatten" type: "Flatten" bottom: "mbox_conf_softmax" top: "mbox_conf_flatten" flatten_param { axis: 1 } }\n'
               'layer { name: "detection_out" type: "DetectionOutput" bottom: "mbox_loc" bottom: "mbox_conf_flatten" bottom: "mbox_priorbox" top: "detection_out"\n'
               '  detection_output_param { num_classes: 2 share_location: true background_label_id: 0\n'
               '    nms_param { nms_threshold: 0.45 top_k: 400 } code_type: CENTER_SIZE keep_top_k: 200 confidence_threshold: 0.01 } }\n')
    return "".join(txt), write_caffemodel(blobs, fp16), blobs


# ---- protobuf wire writer (NetParameter.layer = 100; LayerParameter.name = 1, .blobs = 7; BlobProto.shape = 7,
#      .data = 5 packed, or OpenCV's .raw_data_type = 10 / .raw_data = 12)
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def write_caffemodel(blobs, fp16=False):
    out = bytearray(_ld(1, b"testnet"))
    for name, arrs in blobs.items():
        layer = bytearray(_ld(1, name.encode()) + _ld(2, b"Unknown"))
        for a in arrs:
            shape = _ld(1, b"".join(_varint(int(d)) for d in a.shape))
            if fp16:
                blob = _ld(7, shape) + _varint((10 << 3) | 0) + _varint(2) + _ld(12, a.astype("<f2").tobytes())
            else:
                blob = _ld(7, shape) + _ld(5, a.astype("<f4").tobytes())
            layer += _ld(7, blob)
        out += _ld(100, bytes(layer))
    return bytes(out)
