from oracle.myutils_r import QuantizedTensor, quantize_tensor, dequantize_tensor  # noqa
