// intentionally empty: the reference's main.cu includes a main.h with no content
