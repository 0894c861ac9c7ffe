"""Shared CLI arguments (realtime_codec_agent/utils/cli_utils.py:3-8)."""


def add_common_inference_args(parser):
    parser.add_argument(
        "--llm_model_path",
        default="random:Llama-3.2-1B-magicodec-no-bpe-multi-131k-stereo",
        help="Model directory (config.json + safetensors), an .npz written by llm.save_npz, or 'random:<name>' "
             "for seeded random-init weights of the Llama-3.2-1B codec architecture.",
    )
