"""Constants of the inference path, same names and values as the reference's trace/constants.py (values pinned by
tests/golden/host_functions.json, captured from the reference)."""
CONTROLLER_HEART_BEAT_EXPIRATION = 30
WORKER_HEART_BEAT_INTERVAL = 15
LOGDIR = "./log_dir"

NUM_FRAMES = 8                 # trace/constants.py:6
MAX_FRAMES = 128               # :7  cap applied by process_video (mm_utils.py:430-431)
NUM_FRAMES_PER_SECOND = 1

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"

# placeholder ids spliced by prepare_inputs_labels_for_multimodal (trace/constants.py:47-53)
MMODAL_TOKEN_INDEX = {"IMAGE": -200, "VIDEO": -201, "AUDIO": -202, "TIME": -203, "SCORE": -204, "SYNC": -205}
MMODAL_INDEX_TOKEN = {v: k for k, v in MMODAL_TOKEN_INDEX.items()}
DEFAULT_MMODAL_TOKEN = {"IMAGE": "<image>", "VIDEO": "<video>", "AUDIO": "<audio>", "TIME": "<time>", "SCORE": "<score>",
                        "SYNC": "<sync>"}
