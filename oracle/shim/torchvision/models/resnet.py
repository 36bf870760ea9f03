from oracle.tv042 import (BasicBlock, Bottleneck, ResNet, conv1x1, conv3x3, model_urls,  # noqa
                          resnet18, resnet34, resnet50, resnet101, resnet152)
