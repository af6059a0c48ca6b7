# namespace package root: the product lives in auroralib.compression_amd
