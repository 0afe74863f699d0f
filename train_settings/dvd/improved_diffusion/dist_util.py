from dvd_amd.dist_util import setup_dist, dev, load_state_dict, sync_params, broadcast_blob, shard_documents, rank, world_size  # noqa: F401
