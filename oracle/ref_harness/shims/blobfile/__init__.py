def BlobFile(path, mode="rb"):
    return open(path, mode)
