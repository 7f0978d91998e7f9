"""Prompt templates and keyword lists of the reference (data constants: /root/reference/classic_templates.py,
classic_celeba_templates.py, classic_waterbirds_templates.py) -- one template, two class / two spurious / four group keywords per
dataset; clip_inference.py:35-48 selects them by --dataset."""
TEMPLATES = ["a photo of a {}."]

KEYWORDS = {
    "celeba": {
        "class": ["not blond hair", "blond hair"],
        "spurious": ["female", "male"],
        "group": ["female with not blond hair", "male with not blond hair", "female with blond hair", "male with blond hair"],
    },
    "waterbirds": {
        "class": ["landbird", "waterbird"],
        "spurious": ["land-background", "water-background"],
        "group": ["landbird on land-background", "landbird on water-background", "waterbird on land-background",
                  "waterbird on water-background"],
    },
}


def prompts(dataset, which):
    """the prompt strings in the reference's order (= the column order of the text JSON files)"""
    return [TEMPLATES[0].format(k) for k in KEYWORDS[dataset][which]]
